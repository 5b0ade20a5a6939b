"""Per-kernel averages of every counter of a `rocprofv3 --pmc` run:  python tools/pmc_sum.py <dir> [kernel-name substring]"""
import collections
import csv
import glob
import re
import sys

d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        if pat not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            n[k] += 1
for k, v in sorted(agg.items()):
    print(k, n[k], {c: round(x / n[k]) for c, x in sorted(v.items())})
