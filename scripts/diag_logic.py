"""Diagnostic: runs frames with an RT_DIAG_LOGIC=<round> build (scripts/build_variant.sh dlogic0 -DRT_DIAG_LOGIC=0; RT_HIP_LIB) and
prints how a wave of wf_logic_kernel spends a 64-path chunk of that round (shader clocks between stamps, no waits added)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene(os.environ.get("WORKLOAD", "lambert_1m"), 1)
rs = R.ResidentScene(sc, 0)
rs.render(); rs.sync(); rs.debug_counters(True)
rs.render(); rs.sync()
v = rs.debug_counters(True)
names = ["start -> state machine left", "-> look-ahead ray picked", "-> queue positions (atomics)", "-> trace entries written", "-> classes + ranks (ordered round)",
         "-> end of the chunk"]
tot = v[7]
for n, x in zip(names, v[:6]): print(f"{n:38s} {x / 1e6:10.1f} Mclk  {100.0 * x / max(tot, 1):5.1f} %   {x / max(v[6], 1):9.0f} clk per chunk")
print(f"chunks {v[6]}, {tot / max(v[6], 1):.0f} clk per chunk")
rs.close()
