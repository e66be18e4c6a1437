import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
t0 = int(rows[0]["Start_Timestamp"])
for i in range(len(rows) - 1):
    gap = (int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3
    if gap > 300:
        print(f"at {(int(rows[i]['End_Timestamp']) - t0) / 1e6:9.3f} ms: gap {gap / 1e3:8.3f} ms between {name(rows[i])} and {name(rows[i + 1])}")
