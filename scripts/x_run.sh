export TMPDIR=/tmp
timeout -k 10 300 python3 scripts/x_four.py > gpurun_out/x_four.log 2>&1; tail -60 gpurun_out/x_four.log
