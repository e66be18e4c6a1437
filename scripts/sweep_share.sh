# usage: bash scripts/sweep_share.sh -- one rank's 1/8 tile share (1080p and 4K) under a few segmentation settings
run() { tag=$1; shift; echo "== $tag"; env "$@" python3 scripts/rank_share.py lambert_1m 8 2>/dev/null | tail -1 | cut -c1-160; env "$@" python3 scripts/rank_share.py lambert_4k 8 2>/dev/null | tail -1 | cut -c1-160; }
run default X=1
run seg128 RT_WF_SEG=4096,128,64,16
run seg384 RT_WF_SEG=4096,384,64,16
run seg512_128 RT_WF_SEG=4096,512,128,16
run seg256_96 RT_WF_SEG=4096,256,96,16
run seg256_48 RT_WF_SEG=4096,256,48,16
run rays500k RT_WF_SEG_RAYS=500000,200000,30000
run rays1M RT_WF_SEG_RAYS=1000000,400000,30000
