export TMPDIR=/tmp
out=gpurun_out/r03_newtests
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_frontend_gpu.py tests/test_kat_gpu.py tests/test_builders_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "obj_file or kat or randf or builders or golden or fresh" > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -6 $out/pytest.log
bash scripts/x_multi.sh base
WORKLOAD=lambert_4k bash scripts/x_multi.sh base
