#!/bin/bash
# A/B of the trace input modes (gpurun_out/ab_sort_*.json): RT_WF_SORT = 0 queues, 1 sorted + general kernel, 2 sorted + lean kernel
set -e
mkdir -p gpurun_out
for s in ${MODES:-2 1 0}; do for l in ${LAS:-1}; do
  RT_WF_SORT=$s RT_WF_LOOKAHEAD=$l timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_sort_s${s}_l${l}.json 2> gpurun_out/ab_sort_s${s}_l${l}.err
  python - <<PY
import json
j=json.load(open("gpurun_out/ab_sort_s${s}_l${l}.json"))
print("sort=$s lookahead=$l ms/frame=%.3f stages=%s rounds=%s" % (j["ms_per_step"], j["roofline"]["stage_ms_per_frame"], j["roofline"]["rounds"]))
PY
done; done
