"""Summarise rocprofv3 --pmc CSVs: per-kernel mean of each counter for k_extend dispatches."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_extend" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-40s n=%d last=%.6g mean=%.6g" % (k, len(v), v[-1], sum(v) / len(v)))
