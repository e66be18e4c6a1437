# timing-only experiments (results void): stage times of one frame per variant library
for tag in base "$@"; do
  if [ "$tag" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_$tag.so; fi
  echo "== $tag"; timeout -k 10 120 python3 scripts/rank_share.py lambert_1m 1 2>&1 | grep "N=1"
done
