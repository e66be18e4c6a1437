"""Diagnostic: runs frames with an RT_DIAG_LOGIC build (scripts/build_variant.sh dlogic -DRT_DIAG_LOGIC; RT_HIP_LIB) and prints
where a wave of wf_logic_kernel spends its time per 64-path chunk, rounds 0 and 1 (s_memtime ticks at 100 MHz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene(os.environ.get("WORKLOAD", "lambert_1m"), 1)
rs = R.ResidentScene(sc, 0)
rs.render(); rs.sync(); rs.debug_counters(True)
rs.render(); rs.sync()
v = rs.debug_counters(True)
for r in (0, 1):
    ld, sm, st, n = v[4 * r: 4 * r + 4]
    n = max(n, 1)
    print(f"round {r}: {n} chunks; ticks per chunk: loads {ld / n:.0f}, machine {sm / n:.0f}, stores+append {st / n:.0f}")
rs.close()
