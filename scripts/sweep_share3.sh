# usage: bash scripts/sweep_share3.sh -- all rank shares + whole frames under two segment-length settings, two passes
run() { tag=$1; shift; a=$(env "$@" python3 scripts/rank_share.py lambert_1m 1 2 4 8 2>/dev/null | sed 's/.*share in \([0-9.]*\) ms.*/\1/' | tr '\n' ' '); b=$(env "$@" python3 scripts/rank_share.py lambert_4k 1 8 2>/dev/null | sed 's/.*share in \([0-9.]*\) ms.*/\1/' | tr '\n' ' '); echo "$tag 1080p[1,2,4,8] $a 4K[1,8] $b"; }
for pass in 1 2; do
run default X=1
run s384_96 RT_WF_SEG=4096,384,96,16
run s256_96 RT_WF_SEG=4096,256,96,16
run s384_96_24 RT_WF_SEG=4096,384,96,24
done
