export TMPDIR=/tmp
bash scripts/x_multi.sh prev kw prev kw
WORKLOAD=lambert_4k bash scripts/x_multi.sh prev kw
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/x_pytest.log 2>&1; tail -3 gpurun_out/x_pytest.log
