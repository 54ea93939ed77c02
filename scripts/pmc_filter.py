"""Keep only the rows of one kernel of a rocprofv3 counter_collection.csv (the files are tens of MB with every kernel of a
bench run in them; profiles/ tracks what the summaries are made from):
    pmc_filter.py <kernel substring> <in.csv> <out.csv>"""
import csv
import sys

ksub, src, dst = sys.argv[1:4]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.DictWriter(g, fieldnames=r.fieldnames)
    w.writeheader()
    n = 0
    for row in r:
        if ksub in row["Kernel_Name"]:
            w.writerow(row)
            n += 1
print("%s: %d rows of *%s* -> %s" % (src, n, ksub, dst))
