#!/bin/bash
# Bench of the in-tree library with/without the look-ahead ray (gpurun_out/ab_la_*.json)
set -e
mkdir -p gpurun_out
for l in ${LAS:-1 0}; do
  RT_WF_LOOKAHEAD=$l timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_la_$l.json 2> gpurun_out/ab_la_$l.err
  python - <<PY
import json
j=json.load(open("gpurun_out/ab_la_$l.json"))
print("lookahead=$l ms/frame=%.3f stages=%s rounds=%s" % (j["ms_per_step"], j["roofline"]["stage_ms_per_frame"], j["roofline"]["rounds"]))
PY
done
