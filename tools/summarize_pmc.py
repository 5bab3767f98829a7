#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files into one per-kernel summary
(mean counter value per dispatch).  FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B;
on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)."""
import collections
import csv
import sys


def main():
    out = collections.OrderedDict()
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            key = (r["Kernel_Name"], r["Counter_Name"])
            out.setdefault(key, []).append(float(r["Counter_Value"]))
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "MeanValue"])
    for (k, c), v in out.items():
        w.writerow([k, c, len(v), round(sum(v) / len(v), 2)])


if __name__ == "__main__":
    main()
