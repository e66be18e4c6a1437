export TMPDIR=/tmp
bash scripts/x_multi.sh prev mi prev mi prev mi
WORKLOAD=lambert_4k bash scripts/x_multi.sh prev mi
