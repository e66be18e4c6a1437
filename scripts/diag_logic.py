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
names = ["start -> state words here", "-> ring / resolved hit here", "-> material id here", "-> material + normal", "-> texels, draws, spawns",
         "-> machine left", "-> look-ahead ring entry read", "-> appended, stores issued"]
tot = sum(v)
for n, x in zip(names, v): print(f"{n:34s} {x / 1e6:10.1f} Mclk  {100.0 * x / max(tot, 1):5.1f} %")
rs.close()
