# full GPU test suite + headline bench; logs under gpurun_out/r03_check
export TMPDIR=/tmp
out=gpurun_out/r03_check
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -15 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log && timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $out/bench_lambert1m.json 2> $out/bench_lambert1m.err
python3 - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03_check/bench_lambert1m.json').read().strip().splitlines()[-1])
    print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['stage_ms_per_frame'], d.get('parity'))
except Exception as e: print('no bench line', e)
PY
