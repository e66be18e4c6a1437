"""Summarises the rocprofv3 --pmc CSVs of scripts/pmc_r02.sh: per kernel, the SUM of each counter over its dispatches and
the dispatch count (bench.py passes run 3 timed/warm-up frames + 3-10 stage frames + 1; per-frame = sum / frames, frames =
dispatches of wf_primary_kernel).  The calibration binary's kernels are listed the same way."""
import collections, csv, glob, os, sys
root = sys.argv[1]
for part in ("calib", "bench"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in sorted(glob.glob(os.path.join(root, part + "_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "?").split("(")[0].replace("void ", "")
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    if not acc:
        continue
    print(f"== {part} ==")
    frames = max((cnt.get("wf_primary_kernel", {}) or {"x": 1}).values()) if part == "bench" else 1
    for k in sorted(acc):
        if part == "bench" and not (k.startswith("wf_") or k.startswith("rt_")):
            continue
        print(k)
        for c in sorted(acc[k]):
            n = cnt[k][c]
            per = acc[k][c] / (frames if part == "bench" else n)
            print(f"  {c:38s} sum {acc[k][c]:.6g}  dispatches {n}  {'per frame' if part == 'bench' else 'per dispatch'} {per:.6g}")
for log in sorted(glob.glob(os.path.join(root, "calib_plain.log"))):
    print("== calibration timings ==")
    print(open(log).read())
