# usage: bash scripts/pmc_r02.sh <tag> [workload] [calib]
# Round-2 counter passes (separate rocprofv3 --pmc runs, <= 4 TCC / 8 SQ / 2 GRBM counters each, program directly after --):
#   calibration binary (scripts/micro/bin/gather_calib) and bench.py on `workload`.  Summaries: gpurun_out/pmc2_<tag>/summary*.txt
set -e
tag=$1; wl=${2:-lambert_1m}; calib=${3:-1}
export TMPDIR=/tmp
out=gpurun_out/pmc2_$tag
mkdir -p $out
bench() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $out/bench_$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_$name.log 2>&1 || echo "bench pass $name failed"; }
cal() { name=$1; shift; timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d $out/calib_$name -- scripts/micro/bin/gather_calib > $out/calib_$name.log 2>&1 || echo "calib pass $name failed"; }
if [ "$calib" = "1" ]; then
  scripts/micro/bin/gather_calib > $out/calib_plain.log 2>&1 || echo "calib plain failed"
  cal ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  cal hit TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
  cal fetch FETCH_SIZE
fi
bench ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
bench hit TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
bench fetch FETCH_SIZE
bench write WRITE_SIZE
bench sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
bench sq2 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA
bench ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
bench tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
bench grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
python3 scripts/pmc_summary_r02.py $out > $out/summary.txt
cat $out/summary.txt
