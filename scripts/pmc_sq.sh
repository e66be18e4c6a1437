# usage: bash scripts/pmc_sq.sh <tag> [workload] -- the two SQ counter passes only
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcsq_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
python3 scripts/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
