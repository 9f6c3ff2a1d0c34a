"""Reduces a rocprofv3 counter_collection.csv (one row per dispatch and counter) to one row per kernel and
counter: the mean value and the number of dispatches.  The per-dispatch files of a run with thousands of
launches are tens of megabytes; tools/pmc_traffic.py reads either form.

    python tools/pmc_reduce.py IN.csv OUT.csv"""
import csv
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1:3]
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(src, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            tot[key] += float(row["Counter_Value"])
            cnt[key] += 1
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatches"])
        for key in sorted(tot):
            w.writerow([key[0], key[1], repr(tot[key] / cnt[key]), cnt[key]])


if __name__ == "__main__":
    main()
