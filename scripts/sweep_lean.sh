#!/bin/bash
# usage (on the GPU box): bash scripts/sweep_lean.sh tag1 tag2 ...  -- times each prebuilt variant lib (scripts/build_variant.sh)
for v in "$@"; do
  RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$v.so timeout -k 10 200 python bench.py --workload ${WORKLOAD:-lambert_1m} --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/frame', d['ms_per_step'], 'stages', d['roofline']['stage_ms_per_frame'])"
done
