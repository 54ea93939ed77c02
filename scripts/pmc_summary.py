"""Fold rocprofv3 --pmc counter_collection.csv files into per-launch figures for one kernel.
    pmc_summary.py <kernel substring> <csv> [<csv> ...]
ALL launches of the kernel with its largest grid (the full-size launches: the prepare-launches of one workgroup are
dropped, nothing else) are reported: duration min / median / max under the profiler, and per counter min / median / max.
When GRBM_GUI_ACTIVE is among the counters (it is summed over the 8 XCDs) every launch also gets its shader clock =
GRBM_GUI_ACTIVE / 8 / duration and its cost in shader cycles = duration x clock = GRBM_GUI_ACTIVE / 8: a kernel at its
instruction-issue floor costs the same cycles whatever the clock the power cap leaves."""
import csv
import sys
from collections import defaultdict


def med(v):
    v = sorted(v)
    k = len(v)
    return v[k // 2] if k % 2 else 0.5 * (v[k // 2 - 1] + v[k // 2])


def main():
    ksub = sys.argv[1]
    for path in sys.argv[2:]:
        per = defaultdict(dict)
        dur, grid, meta = {}, {}, {}
        for r in csv.DictReader(open(path)):
            if ksub not in r["Kernel_Name"]:
                continue
            d = r["Dispatch_Id"]
            per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            grid[d] = int(r["Grid_Size"])
            meta[d] = {"vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"],
                       "lds": r["LDS_Block_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"],
                       "name": r["Kernel_Name"].split("(")[0][-60:]}
        if not per:
            continue
        gmax = max(grid.values())
        keep = sorted((d for d in per if grid[d] == gmax), key=int)
        ds = [dur[d] for d in keep]
        print("# %s: %d launches of *%s* at the full grid (%d in all); %s" % (path, len(keep), ksub, len(per), meta[keep[0]]))
        print("  duration under the profiler [us]: min %.1f  median %.1f  max %.1f" % (min(ds), med(ds), max(ds)))
        if all("GRBM_GUI_ACTIVE" in per[d] for d in keep):
            print("  per launch: duration us, shader clock GHz (GRBM_GUI_ACTIVE / 8 / duration), shader cycles (M)")
            for d in keep:
                cyc = per[d]["GRBM_GUI_ACTIVE"] / 8.0
                print("    %8.1f  %5.3f  %6.3f" % (dur[d], cyc / dur[d] / 1e3, cyc / 1e6))
            cycs = [per[d]["GRBM_GUI_ACTIVE"] / 8.0 / 1e6 for d in keep]
            print("  shader cycles per launch [M]: min %.3f  median %.3f  max %.3f" % (min(cycs), med(cycs), max(cycs)))
        names = sorted({k for d in keep for k in per[d]})
        print("  %-28s %14s %14s %14s" % ("counter (per launch)", "min", "median", "max"))
        for nme in names:
            vals = [per[d][nme] for d in keep if nme in per[d]]
            print("  %-28s %14.6g %14.6g %14.6g" % (nme, min(vals), med(vals), max(vals)))


if __name__ == "__main__":
    main()
