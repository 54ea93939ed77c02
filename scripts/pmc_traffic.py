"""HBM bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md's HBM section prescribes for gfx950 (FETCH_SIZE counts 128-B requests as 64 B: x2; both
counters are in KiB).  Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>
<kernel substring> <m> <n> <label> <pivots_per_sweep> [<steps of the profiled command>]  -> JSON on stdout (profiles/traffic_<workload>_n1.json;
bench.py quotes it as roofline.traffic only when workload, GPU count and pivots per sweep match its own run).

Only FULL launches are averaged: those whose counter value is within 5 % of the largest one (the warm-up and the
last block of a budgeted run apply fewer pivots or run other template instances)."""
import csv
import json
import sys


def per_launch(path, counter, kernel_sub):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or kernel_sub not in r["Kernel_Name"]:
            continue
        key = r["Dispatch_Id"]
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    vals = list(acc.values())
    top = max(vals) if vals else 0.0
    return [v for v in vals if v >= 0.95 * top]   # full launches only (see the module docstring)


def main():
    fpath, wpath, ksub, m, n, label = sys.argv[1:7]
    m, n = int(m), int(n)
    block = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    steps = int(sys.argv[8]) if len(sys.argv) > 8 else None
    f = per_launch(fpath, "FETCH_SIZE", ksub)
    w = per_launch(wpath, "WRITE_SIZE", ksub)
    one_pass = 16.0 * m * n
    fmean = sum(f) / len(f)
    wmean = sum(w) / len(w)
    rd, wr = 2.0 * fmean * 1024.0, wmean * 1024.0
    out = {
        "workload": label,
        "pivots_per_sweep": block,
        "kernel": ksub,
        "steps": steps,
        "date": __import__("datetime").date.today().isoformat(),
        "launches_sampled": [len(f), len(w)],
        "FETCH_SIZE_KiB_mean": fmean,
        "WRITE_SIZE_KiB_mean": wmean,
        "correction": "read bytes = 2*FETCH_SIZE*1024 (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM "
                      "section); write bytes = WRITE_SIZE*1024",
        "hbm_read_bytes_per_launch": rd,
        "hbm_write_bytes_per_launch": wr,
        "hbm_bytes_per_launch": rd + wr,
        "bytes_one_pass_over_the_tableau": one_pass,
        "ratio_traffic_over_one_pass": (rd + wr) / one_pass,
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
