# usage: bash scripts/sweep_l4.sh -- headline frame under several lengths of the fourth segment level (rounds of 30-100 k rays), three passes
run() { tag=$1; shift; env "$@" python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['roofline']['stage_ms_per_frame']['trace'], d['roofline']['stage_ms_per_frame']['sort'])"; }
for pass in 1 2 3; do
run l4_64 X=1
run l4_40 RT_WF_SEG=4096,384,96,40,16
run l4_48 RT_WF_SEG=4096,384,96,48,16
run l4_80 RT_WF_SEG=4096,384,96,80,16
run l4_96 RT_WF_SEG=4096,384,96,96,16
done
