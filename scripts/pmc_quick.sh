# usage: bash scripts/pmc_quick.sh <tag> [workload] -- three counter passes of bench.py (fabric requests, L2 hit/miss, SQ), summary of the wf_ kernels
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcq_$tag
mkdir -p $out
bench() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out/bench_$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$name.log 2>&1 || echo "bench pass $name failed"; }
bench hit TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
bench sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
python3 scripts/pmc_summary_r02.py $out | awk '/^wf_trace|^wf_setup|^wf_scatter/{p=1} /^wf_accum|^wf_logic|^wf_primary|^rt_/{p=0} p' > $out/summary.txt
cat $out/summary.txt
