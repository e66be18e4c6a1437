"""Where the time of RaytraceAll's all-GPUs mode goes on a one-GPU box (four instances of the 1 M-triangle scene)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_TIMING"] = "1"
import bench
from opencl_render_amd import raytrace as R
sc = bench.make_scene("lambert_1m", 1)
n = R.lib().rtHipDeviceCount()
for rep in range(2):
    R.lib().rtHipCacheClear()
    os.environ.pop("RT_HIP_VIRTUAL_DEVICES", None)
    t0 = time.perf_counter(); ok, *_ = R.raytrace_all(1, sc); one = time.perf_counter() - t0
    R.lib().rtHipCacheClear()
    os.environ["RT_HIP_VIRTUAL_DEVICES"] = "4"
    print("---- four", flush=True)
    t0 = time.perf_counter(); ok, *_ = R.raytrace_all(n + 1, sc); four = time.perf_counter() - t0
    print(f"one {one*1e3:.1f} ms, four {four*1e3:.1f} ms", flush=True)
