# per-kernel durations of one planned frame for (library variant, environment) pairs: "tag[:VAR=val,VAR=val]"
export TMPDIR=/tmp
wl=${WORKLOAD:-lambert_1m}; n=${SHARE:-1}
for spec in "$@"; do
  tag=${spec%%:*}; envs=""; [ "$spec" != "$tag" ] && envs=${spec#*:}
  ( if [ "$tag" != base ]; then export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
    IFS=','; for kv in $envs; do export "$kv"; done; unset IFS
    out=gpurun_out/xm_$(echo $spec | tr ':=,' '___'); rm -rf $out; mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/rank_share.py $wl $n > $out/run.log 2>&1
    grep "N=$n" $out/run.log | sed 's/rank 0 renders its share in//;s/(ideal.*stages/stages/'
    python3 - $out "$spec" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "wf_primary" in r["Kernel_Name"]]
s = idx[len(idx)//2]; e = idx[len(idx)//2 + 1]
print("   ", sys.argv[2], " ".join(f'{r["Kernel_Name"].split("(")[0].replace("void ","").replace("wf_","").replace("_kernel","")}={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:.1f}' for r in rows[s:e]))
PY
  )
done
