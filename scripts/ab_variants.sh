# usage: bash scripts/ab_variants.sh <workload> <tag>... -- bench.py (no CPU baseline) with the in-tree library ("base") and with each
# opencl_render_amd/variants/lib_<tag>.so; prints ms/frame and stage times
wl=$1; shift
mkdir -p gpurun_out
for tag in base "$@"; do
  if [ "$tag" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
  python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abv_$tag.json 2> gpurun_out/abv_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/abv_$tag.err; continue; }
  python3 - $tag <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abv_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["roofline"]["stage_ms_per_frame"], "frac", d["roofline"]["frac"])
PY
done
