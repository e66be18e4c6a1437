"""Diagnostic: runs frames with an RT_DIAG_STAMPS build (RT_HIP_LIB) and prints the sorted trace kernel's cycle anatomy.
Build:  hipcc <DEVFLAGS of csrc/Makefile> -DRT_DIAG_STAMPS -c rt_wavefront.hip -o build/wf_diag.o  and link it into
opencl_render_amd/variants/lib_diag.so in place of rt_wavefront.o (scripts/build_variant.sh diag -DRT_DIAG_STAMPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene(os.environ.get("WORKLOAD", "lambert_1m"), 1)
rs = R.ResidentScene(sc, 0)
rs.render(); rs.sync(); rs.debug_counters(True)
rs.render(); rs.sync()
total, walk, test, witers, batches, steps, waves, items = rs.debug_counters(True)
unroll = int(os.environ.get("UNROLL", "5"))  # RT_WF_BLIND
print(f"waves {waves}: cycles/wave {total / waves:.0f} = walk {walk / waves:.0f} ({witers / waves:.1f} loop heads x {unroll} steps, "
      f"{walk / max(witers * unroll, 1):.0f} cyc per step) "
      f"+ test {test / waves:.0f} ({batches / waves:.1f} batches, {test / max(batches, 1):.0f} cyc each, {items / max(batches, 1):.1f} items per batch) "
      f"+ rest {(total - walk - test) / waves:.0f};  before the walk (staging, planning, cuts) {steps / waves:.0f} per wave pass")
rs.close()
