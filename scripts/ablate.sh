# usage: bash scripts/ablate.sh v1 v2 ... -- per-launch times of the first round for ablation variants (RESULTS ARE WRONG by design)
export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/abl_$v; mkdir -p $out
  RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$v.so RT_WF_BUDGETS="64,128" timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
  python3 - $out $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
start = idx[1] if len(idx) > 1 else idx[0]
out = []
for r in rows[start:start + 6]:
    out.append(f'{r["Kernel_Name"].split("(")[0].replace("void ","")[3:12]}={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.0f}us')
print(sys.argv[2], " ".join(out))
PY
done
