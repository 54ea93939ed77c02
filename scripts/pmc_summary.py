"""Fold rocprofv3 --pmc counter_collection.csv files into per-launch means for one kernel (largest launches only:
the full blocks).  Usage: pmc_summary.py <kernel substring> <csv> [<csv> ...]"""
import csv
import sys
from collections import defaultdict


def main():
    ksub = sys.argv[1]
    for path in sys.argv[2:]:
        per = defaultdict(dict)
        dur = {}
        meta = {}
        for r in csv.DictReader(open(path)):
            if ksub not in r["Kernel_Name"]:
                continue
            d = r["Dispatch_Id"]
            per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            meta = {"vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"],
                    "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]}
        if not per:
            continue
        dmax = max(dur.values())
        keep = [d for d in per if dur[d] > 0.8 * dmax]
        print("# %s: %d launches of *%s* (of %d) within 20%% of the longest (%.1f us); %s" % (path.split("/")[-3], len(keep), ksub, len(per), dmax, meta))
        names = sorted({k for d in keep for k in per[d]})
        print("  mean duration under the profiler: %.1f us" % (sum(dur[d] for d in keep) / len(keep)))
        for nme in names:
            vals = [per[d][nme] for d in keep if nme in per[d]]
            print("  %-28s %.6g" % (nme, sum(vals) / len(vals)))


if __name__ == "__main__":
    main()
