# usage: bash scripts/pmc_ifetch.sh <tag> [workload] -- instruction-fetch and wait counters of the frame's kernels
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcif_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || echo "pass $name failed"; }
run if1 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
run if2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS
python3 scripts/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
