# usage: bash scripts/ktrace.sh <tag> [workload] -- per-dispatch kernel trace (one frame's launches listed in order)
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/ktrace_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void wf_", "wf_", "rt_", "void rt_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print the last complete frame: from the last wf_primary to the following wf_accum
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
start = idx[1] if len(idx) > 1 else idx[0]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f}us  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f}us  {name}')
    if "wf_accum" in name:
        break
PY
