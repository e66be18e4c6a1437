# usage: bash scripts/ab_variants.sh v1 v2 ...   (variants = opencl_render_amd/variants/lib_<v>.so), optional WORKLOAD env
set -e
for v in "$@"; do
  RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$v.so timeout -k 10 200 python bench.py --workload ${WORKLOAD:-lambert_1m} --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '${WORKLOAD:-lambert_1m}', 'ms/frame', d['ms_per_step'], 'stages', d['roofline']['stage_ms_per_frame'])"
done
