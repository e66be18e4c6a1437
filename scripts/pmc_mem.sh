# usage: bash scripts/pmc_mem.sh <tag> [workload] -- memory-pipeline counters (TA / TCP / TCC / GRBM), separate passes
set -e
tag=$1; wl=${2:-lambert_1m}
export TMPDIR=/tmp
out=gpurun_out/pmcmem_$tag
mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || echo "pass $name failed"; }
run ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tcc1 TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum
run tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_BUSY_avr
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
python3 scripts/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
