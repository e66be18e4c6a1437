# round 3, first GPU call: the new tests, the bench line at HEAD, rank shares for the configs round 2 left out
export TMPDIR=/tmp
out=gpurun_out/r03_first
mkdir -p $out
python3 -m pytest tests/test_rccl_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "rccl or tail or more_lights or cache" > $out/pytest_new.log 2>&1; echo "pytest rc $?" >> $out/pytest_new.log
tail -5 $out/pytest_new.log
python3 bench.py --steps 20 --warmup 3 > $out/bench_lambert1m.json 2> $out/bench_lambert1m.err
tail -c 600 $out/bench_lambert1m.json
python3 scripts/rank_share.py lambert_1m --samples 4 1 2 4 8 > $out/rank_share_1m_s4.txt 2>&1
python3 scripts/rank_share.py lambert_1m --samples 16 1 8 > $out/rank_share_1m_s16.txt 2>&1
python3 scripts/rank_share.py lambert_4k --samples 4 1 8 > $out/rank_share_4k_s4.txt 2>&1
python3 scripts/rank_share.py lambert_10m_4k 1 2 4 8 > $out/rank_share_10m_4k.txt 2>&1
cat $out/rank_share_*.txt | grep "N="
