export TMPDIR=/tmp
bash scripts/x_multi.sh prev base early prev base early
SHARE=8 bash scripts/x_multi.sh prev base early
