for r in 0 1; do echo "== round $r"; RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_dlogic$r.so python3 scripts/diag_logic.py 2>/dev/null | tail -8; done
