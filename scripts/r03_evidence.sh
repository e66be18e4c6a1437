# usage: bash scripts/r03_evidence.sh -- everything profiles/r03_* is made from, in one GPU call:
#   bench lines of the workloads (+ S=4, S=16), rocprofv3 --kernel-trace --stats of the headline bench command, counter passes (fabric
#   requests by size, L2 hit/miss, writes, SQ, cycles) on lambert_1m, SQ counters of the primary kernel on lambert_4k, the share of one of
#   1/2/4/8 ranks for every BASELINE config, and the RCCL world-1 test's log.
export TMPDIR=/tmp
out=gpurun_out/r03_evidence
mkdir -p $out
python3 bench.py --steps 20 --warmup 3 > $out/bench_lambert1m.json 2> $out/bench_lambert1m.err; echo "bench lambert_1m rc $?"
for wl in primary_100k lambert_4k lambert_10m_4k; do
  python3 bench.py --workload $wl --steps 10 --warmup 2 > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl rc $?"
done
python3 bench.py --samples 4 --steps 10 --warmup 2 > $out/bench_s4.json 2> $out/bench_s4.err; echo "bench S=4 rc $?"
python3 bench.py --samples 16 --steps 6 --warmup 2 > $out/bench_s16.json 2> $out/bench_s16.err; echo "bench S=16 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/ktrace.log 2>&1
pass() { name=$1; wl=$2; shift; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$name.log 2>&1 || echo "pass $name failed"; }
pass ea lambert_1m TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass hit lambert_1m TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
pass write lambert_1m WRITE_SIZE
pass sq lambert_1m SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pass cyc lambert_1m GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq4k lambert_4k SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pass grbm4k lambert_4k GRBM_GUI_ACTIVE GRBM_COUNT
echo "counter passes done"
python3 scripts/rank_share.py lambert_1m 1 2 4 8 > $out/rank_share_1m.txt 2>&1
python3 scripts/rank_share.py lambert_1m --samples 4 1 8 > $out/rank_share_1m_s4.txt 2>&1
python3 scripts/rank_share.py lambert_1m --samples 16 1 8 > $out/rank_share_1m_s16.txt 2>&1
python3 scripts/rank_share.py lambert_4k 1 2 4 8 > $out/rank_share_4k.txt 2>&1
python3 scripts/rank_share.py lambert_4k --samples 4 1 8 > $out/rank_share_4k_s4.txt 2>&1
python3 scripts/rank_share.py lambert_10m_4k 1 2 4 8 > $out/rank_share_10m_4k.txt 2>&1
grep -h "N=" $out/rank_share_*.txt
timeout -k 10 300 python3 -m pytest tests/test_rccl_gpu.py -v -m gpu > $out/rccl_world1.log 2>&1; tail -3 $out/rccl_world1.log
ls $out
