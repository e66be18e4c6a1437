# usage: bash scripts/sweep_seg2.sh [workload] -- bench.py over a few RT_WF_SEG / RT_WF_SEG_RAYS / RT_WF_APPEND_RAYS settings (no CPU baseline)
wl=${1:-lambert_1m}
run() { tag=$1; shift; env "$@" python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['roofline']['stage_ms_per_frame'])"; }
run default X=1
run seg24 RT_WF_SEG=4096,256,64,24
run seg32 RT_WF_SEG=4096,256,64,32
run seg48 RT_WF_SEG=4096,256,64,48
run seg12 RT_WF_SEG=4096,256,64,12
run r60k RT_WF_SEG_RAYS=700000,300000,60000
run r100k_32 RT_WF_SEG_RAYS=700000,300000,100000 RT_WF_SEG=4096,256,48,24
run app80k RT_WF_APPEND_RAYS=80000
run app0 RT_WF_APPEND_RAYS=0
