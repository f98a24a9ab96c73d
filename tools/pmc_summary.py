"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel name."""
import csv, sys, collections, glob, os
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name") or row.get("Kernel Name")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("kernel,counter,launches,mean,total")
for k, cs in sorted(acc.items()):
    for c, v in cs.items():
        print(f'"{k[:90]}",{c},{len(v)},{sum(v)/len(v):.1f},{sum(v):.1f}')
