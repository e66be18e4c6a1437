export TMPDIR=/tmp
bash scripts/run_diag_trace.sh
