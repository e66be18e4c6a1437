#!/bin/bash
# A/B of the number of concurrent tile groups per instance (RT_WF_GROUPS)
for g in ${GROUPS_LIST:-1 2 3 4}; do
  RT_WF_GROUPS=$g timeout -k 10 200 python bench.py --workload ${WORKLOAD:-lambert_1m} --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('groups $g', 'ms/frame', d['ms_per_step'], 'Mrays/s', d['value'], 'stages', d['roofline']['stage_ms_per_frame'])"
done
