# usage: bash scripts/ab_trace.sh <tag>... -- trace-kernel time per frame (HIP events, 40 frames) of the in-tree library and variants, three rounds interleaved
for r in 1 2 3; do
for tag in base "$@"; do
  if [ "$tag" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['roofline']['stage_ms_per_frame']['trace'])"
done; done
