# usage: bash scripts/xlogic.sh <tag>... -- per-launch durations of one frame's kernels (rocprofv3 --kernel-trace) for the in-tree library ("base") and variants lib_<tag>.so
export TMPDIR=/tmp
for tag in "$@"; do
  out=gpurun_out/xl_$tag; mkdir -p $out
  if [ "$tag" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --workload lambert_1m --steps 2 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
  python3 - $out $tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void wf_", "wf_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
start = idx[-1]
print(sys.argv[2], " ".join(f'{r["Kernel_Name"].split("(")[0][3:-7]}={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.1f}' for r in rows[start:start + 12]))
PY
done
