export TMPDIR=/tmp
timeout -k 10 600 python3 scripts/x_soak.py > gpurun_out/x_soak.log 2>&1; tail -8 gpurun_out/x_soak.log
