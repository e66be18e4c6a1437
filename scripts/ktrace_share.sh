# usage: bash scripts/ktrace_share.sh <tag> <N> [workload] -- per-dispatch kernel trace of rank 0's tile share of an N-rank run
set -e
tag=$1; n=$2; wl=${3:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/ktshare_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/rank_share.py $wl $n > $out/run.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void wf_", "wf_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
start = idx[10]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:idx[11]]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f}us  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f}us  {name}')
PY
