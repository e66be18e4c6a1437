# quick GPU check: parity subset, diag stamps, bench
export TMPDIR=/tmp
out=gpurun_out/r03_quick
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden or modes or region_b or lights or planned or batches or fresh" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -4 $out/pytest.log

timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_quick/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_frame'])
PY
