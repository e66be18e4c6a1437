"""Experiment: ms/frame of successive groups of 10 frames from a cold start (does the chip speed up as it stays busy?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene(sys.argv[1] if len(sys.argv) > 1 else "lambert_1m", 1)
rs = R.ResidentScene(sc, 0, None)
st = torch.cuda.Stream(torch.device("cuda", 0))
for _ in range(3):
    rs.render(st.cuda_stream)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "0.5")))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record(st)
for g in range(40):
    for _ in range(10):
        rs.render(st.cuda_stream)
    ev[g + 1].record(st)
torch.cuda.synchronize()
print(" ".join(f"{ev[g].elapsed_time(ev[g + 1]) / 10:.3f}" for g in range(40)))
print("redone", rs.finish())
rs.close()
