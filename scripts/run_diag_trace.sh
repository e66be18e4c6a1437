for r in 1 2; do echo "== trace round $r"; RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_dst$r.so python3 scripts/diag_stamps.py | tail -2; done
