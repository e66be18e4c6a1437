"""Diagnostic: runs one frame with an RT_DIAG_STAMPS build (RT_HIP_LIB) and prints the trace kernel's cycle anatomy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene("lambert_1m", 1)
rs = R.ResidentScene(sc, 0)
rs.render(); rs.sync(); rs.debug_counters(True)
rs.render(); rs.sync()
total, walk, test, witers, batches, tlanes, waves, cells = rs.debug_counters(True)
print(f"waves {waves}: cycles/wave {total / waves:.0f} = walk {walk / waves:.0f} ({witers / waves:.1f} iterations, {walk / max(witers, 1):.0f} cyc each) "
      f"+ test {test / waves:.0f} ({batches / waves:.1f} batches, {test / max(batches, 1):.0f} cyc each, {tlanes / max(batches, 1):.1f} lanes and "
      f"{cells / max(batches, 1):.1f} cells per batch) + rest {(total - walk - test) / waves:.0f}")
rs.close()
