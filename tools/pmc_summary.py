"""Summarise rocprofv3 --pmc CSV output (one row per dispatch and counter) per kernel: mean per launch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kern, ctr), (s, n) in sorted(acc.items()):
    print("%-50s %-24s launches=%-5d mean=%.6g" % (kern, ctr, n, s / n))
