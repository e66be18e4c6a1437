# per-kernel durations (rocprofv3 kernel trace) of one planned frame, per variant library
export TMPDIR=/tmp
for tag in base "$@"; do
  if [ "$tag" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
  out=gpurun_out/xk_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/rank_share.py lambert_1m 1 > $out/run.log 2>&1
  python3 - $out $tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
s = idx[len(idx)//2]; e = idx[len(idx)//2 + 1]
print(sys.argv[2], " ".join(f'{r["Kernel_Name"].split("(")[0].replace("void ","").replace("wf_","").replace("_kernel","")}={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.1f}' for r in rows[s:e]))
PY
done
