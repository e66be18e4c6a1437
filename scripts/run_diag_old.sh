cd scratch_old
for r in 1 2; do echo "== OLD trace round $r"; RT_HIP_LIB=$PWD/opencl_render_amd/variants/lib_dst$r.so python3 scripts/diag_stamps.py 2>&1 | tail -1; done
cd ..
bash scripts/run_diag_trace.sh
