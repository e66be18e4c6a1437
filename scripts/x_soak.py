"""Soak: thousands of planned frames back to back (the control words are only ever zeroed by wf_status_kernel), planes compared with the
watched first frame at intervals; then the same for a golden scene with deep paths and for a two-sample-batch frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bench
from opencl_render_amd import raytrace as R, scene as S


def soak(sc, frames, every, what):
    rs = R.ResidentScene(sc, 0)
    rs.render(); first = [p.copy() for p in rs.readback()]
    t0 = time.perf_counter()
    redone = 0
    for i in range(frames):
        rs.render()
        if (i + 1) % every == 0:
            redone += int(rs.finish())
            now = rs.readback()
            assert all(np.array_equal(a, b) for a, b in zip(first, now)), f"{what}: frame {i} differs from the first"
    rs.sync()
    print(f"{what}: {frames} frames, {1e3 * (time.perf_counter() - t0) / frames:.3f} ms/frame incl. {frames // every} readbacks, frames redone {redone}", flush=True)
    rs.close()


soak(bench.make_scene("lambert_1m", 1), 3000, 500, "lambert_1m S=1")
soak(bench.make_scene("lambert_1m", 4), 300, 100, "lambert_1m S=4")
from conftest import load_golden_scene
for name in ("mirror_hall", "all_light_types"):
    try:
        sc, _ = load_golden_scene(name)
    except Exception as e:
        print("no golden", name, e); continue
    soak(sc, 2000, 400, name)
os.environ["RT_WF_STATE_MB"] = "1"
sc = S.make_soup(320, 200, 5000, 0.05, seed=5, samples=6, name="six samples in small batches")
R.build_lists(sc)
soak(sc, 500, 100, "320x200 S=6 in batches")
