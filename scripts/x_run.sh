export TMPDIR=/tmp
bash scripts/x_multi.sh base cs cs2 base cs cs2
WORKLOAD=lambert_4k bash scripts/x_multi.sh base cs cs2
