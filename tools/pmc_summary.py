"""Condense a rocprofv3 `--pmc ... --output-format csv` counter_collection.csv into one row per kernel
(sum of every counter over all dispatches of that kernel).

usage: python tools/pmc_summary.py <counter_collection.csv> > profiles/rNN_pmc_<what>.csv"""
import csv, sys, collections

rows = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
counters = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
    counters.add(r["Counter_Name"])
counters = sorted(counters)
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Dispatches"] + counters)
for k in sorted(rows, key=lambda k: -rows[k][counters[0]]):
    w.writerow([k, len(disp[k])] + [rows[k][c] for c in counters])
