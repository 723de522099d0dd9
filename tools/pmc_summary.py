"""Average rocprofv3 --pmc counter values per dispatch for kernels whose name contains a substring.
usage: python tools/pmc_summary.py <dir with pass*/...counter_collection.csv> <kernel-substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value (summed over dimensions)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if pat not in row.get("Kernel_Name", ""):
                continue
            acc[row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
for name in sorted(acc):
    v = list(acc[name].values())
    print(f"{name:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
