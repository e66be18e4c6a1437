export TMPDIR=/tmp
run() { python3 scripts/rank_share.py lambert_1m $1 2>&1 | grep "N=$1" | sed 's/rank 0 renders its share in//;s/(ideal.*stages/stages/' | cut -c1-150; }
for seg in "4096,384,96,64,16" "4096,384,64,64,16" "4096,384,128,64,16" "4096,384,160,64,16" "4096,384,96,48,16" "4096,384,96,96,16" "4096,256,96,64,16" "4096,512,96,64,16"; do
  export RT_WF_SEG=$seg; echo "== seg $seg"
  run 8; run 4; run 2
done
