"""Summarises rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        if "rt_" not in k and "wf_" not in k:
            continue
        acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean/dispatch {sum(v)/len(v):.4g}  (n={len(v)})")
