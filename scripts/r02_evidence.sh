# usage: bash scripts/r02_evidence.sh -- everything profiles/r02_* is made from, in one GPU call:
#   bench lines of the five workloads, rocprofv3 --kernel-trace --stats of the headline bench command, counter passes
#   (fabric requests, L2 hit/miss, SQ) on lambert_1m, SQ counters of the primary kernel on lambert_4k, rank shares.
export TMPDIR=/tmp
out=gpurun_out/r02_evidence
mkdir -p $out
python3 bench.py --steps 20 --warmup 3 > $out/bench_lambert1m.json 2> $out/bench_lambert1m.err
for wl in primary_100k lambert_4k lambert_10m_4k; do
  python3 bench.py --workload $wl --steps 10 --warmup 2 > $out/bench_$wl.json 2> $out/bench_$wl.err
done
python3 bench.py --samples 4 --steps 10 --warmup 2 > $out/bench_s4.json 2> $out/bench_s4.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/ktrace.log 2>&1
pass() { name=$1; wl=$2; shift; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$name.log 2>&1 || echo "pass $name failed"; }
pass ea lambert_1m TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass hit lambert_1m TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum
pass write lambert_1m WRITE_SIZE
pass sq lambert_1m SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pass sq4k lambert_4k SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pass grbm4k lambert_4k GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/rank_share.py lambert_1m 1 2 4 8 > $out/rank_share_1m.txt 2>&1
python3 scripts/rank_share.py lambert_4k 1 8 > $out/rank_share_4k.txt 2>&1
bash scripts/pmc_valu.sh evidence > $out/valu_busy.txt 2>&1; cp gpurun_out/pmcv_evidence/valu.json $out/valu_busy.json
ls $out
