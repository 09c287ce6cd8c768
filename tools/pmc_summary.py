"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (developer tool)."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, d in agg.items():
    n = len(cnt[k])
    print(f"== {k}  ({n} dispatches)")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / n:16.1f}")
