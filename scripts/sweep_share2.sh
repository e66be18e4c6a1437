# usage: bash scripts/sweep_share2.sh -- the 1/8 tile shares (1080p, 4K) under several segment lengths, three interleaved passes
run() { tag=$1; shift; a=$(env "$@" python3 scripts/rank_share.py lambert_1m 8 2>/dev/null | tail -1 | sed 's/.*share in \([0-9.]*\) ms.*/\1/'); b=$(env "$@" python3 scripts/rank_share.py lambert_4k 8 2>/dev/null | tail -1 | sed 's/.*share in \([0-9.]*\) ms.*/\1/'); echo "$tag 1080p $a 4K $b"; }
for pass in 1 2 3; do
run default X=1
run l3_48 RT_WF_SEG=4096,256,48,16
run l3_96 RT_WF_SEG=4096,256,96,16
run l3_128 RT_WF_SEG=4096,256,128,16
run l3_192 RT_WF_SEG=4096,256,192,16
run l2_160 RT_WF_SEG=4096,160,64,16
run l2_384 RT_WF_SEG=4096,384,64,16
done
