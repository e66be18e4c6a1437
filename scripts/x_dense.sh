export TMPDIR=/tmp
bash scripts/x_multi.sh base b4 b6 base b4 b6
timeout -k 10 600 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/x_dense_pytest.log 2>&1; tail -3 gpurun_out/x_dense_pytest.log
