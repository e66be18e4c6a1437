# usage: VAR=RT_WF_BUDGETS bash scripts/ab_env.sh "val1" "val2" ...  -- bench the in-tree library under different env settings
set -e
for v in "$@"; do
  env ${VAR:-RT_WF_BUDGETS}="$v" timeout -k 10 200 python bench.py --workload ${WORKLOAD:-lambert_1m} --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${VAR:-RT_WF_BUDGETS}=$v', 'ms/frame', d['ms_per_step'], d['roofline']['stage_ms_per_frame'])"
done
