"""Experiment: one frame rendered as K independent tile subsets on K streams of ONE GPU (threads), vs one subset."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from opencl_render_amd import raytrace as R

sc = bench.make_scene(os.environ.get("WORKLOAD", "lambert_1m"), 1)
ref = None
for K in [1, 2, 3, 4, 6, 8]:
    parts = [R.ResidentScene(sc, 0, R.tiles_of_rank(sc.width, sc.height, k, K)) for k in range(K)]
    def frame():
        ts = [threading.Thread(target=lambda p=p: (p.render(), p.sync())) for p in parts]
        for t in ts: t.start()
        for t in ts: t.join()
    for _ in range(3): frame()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n): frame()
    dt = (time.perf_counter() - t0) / n
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    for p in parts: p.readback(planes)
    if ref is None: ref = [x.copy() for x in planes]
    same = all(np.array_equal(a, b) for a, b in zip(ref, planes))
    print(f"K={K}: {dt * 1e3:.3f} ms/frame  identical={same}", flush=True)
    for p in parts: p.close()
