# usage: bash scripts/ab_region.sh <tag> [workload] -- bench.py with region-ordered tracing off and on (no CPU baseline), stage times in the JSON
set -e
tag=$1; wl=${2:-lambert_1m}
mkdir -p gpurun_out
RT_WF_REGION_RAYS=4000000000 python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_region_${tag}_off.json 2> gpurun_out/ab_region_${tag}_off.err
RT_WF_REGION_RAYS=300000 python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_region_${tag}_on.json 2> gpurun_out/ab_region_${tag}_on.err
python3 - <<PY
import json
for k in ("off", "on"):
    d = json.loads(open("gpurun_out/ab_region_${tag}_%s.json" % k).read().strip().splitlines()[-1])
    print(k, d["ms_per_step"], d["roofline"]["stage_ms_per_frame"], d["roofline"]["frac"])
PY
