export TMPDIR=/tmp
out=gpurun_out/x_overlap
mkdir -p $out
timeout -k 10 300 python3 scripts/x_overlap.py lambert_1m > $out/a.log 2>&1 && timeout -k 10 300 python3 scripts/x_overlap.py lambert_4k > $out/b.log 2>&1 && timeout -k 10 300 python3 scripts/x_overlap.py lambert_1m --samples 4 > $out/c.log 2>&1
grep -h "in flight\|same planes" $out/*.log
