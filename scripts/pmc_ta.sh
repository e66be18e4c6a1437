# usage: bash scripts/pmc_ta.sh <tag> [workload] -- vector-L1 (TCP) access counts and texture-addresser busy cycles per kernel
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcta_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || echo "pass $name failed"; }
run ta1 TA_BUSY_avr TA_BUSY_max TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run tcp1 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES
python3 scripts/pmc_summary.py $out > $out/summary.txt
grep -B1 -A14 "wf_trace_sorted" $out/summary.txt
