# sweep of the not-ordered rounds' parameters on lambert_1m: rays per workgroup x segment length (per-kernel times from a kernel trace)
export TMPDIR=/tmp
for g in 32 64; do for sg in 32 48 64 96; do
  export RT_WF_GROUP_RAYS=$g RT_WF_SEG=4096,384,96,$sg,16
  out=gpurun_out/sw2_${g}_$sg; rm -rf $out; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/rank_share.py lambert_1m 1 > $out/run.log 2>&1
  python3 - $out "R=$g seg=$sg" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
s = idx[len(idx)//2]; e = idx[len(idx)//2 + 1]
print(sys.argv[2], " ".join(f'{r["Kernel_Name"].split("(")[0].replace("void ","").replace("wf_","").replace("_kernel","")}={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.1f}' for r in rows[s:e]))
PY
done; done
