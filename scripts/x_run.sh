export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "baseline_config" --durations=5 > gpurun_out/x_pytest.log 2>&1; tail -12 gpurun_out/x_pytest.log
timeout -k 10 300 python3 bench.py --workload lambert_4k --steps 5 --warmup 2 > gpurun_out/x_b4k.json 2> gpurun_out/x_b4k.err; python3 -c "
import json;d=json.loads(open('gpurun_out/x_b4k.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],d['parity'],d['cpu_baseline'])"
