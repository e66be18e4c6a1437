"""Times one rank's share of the frame on one GPU: what a rank of an N-GPU run does per frame, without the gather.
usage: python scripts/rank_share.py [workload] [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
wl = sys.argv[1] if len(sys.argv) > 1 else "lambert_1m"
ns = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
sc = bench.make_scene(wl, 1)
for n in ns:
    rs = R.ResidentScene(sc, 0, R.tiles_of_rank(sc.width, sc.height, 0, n) if n > 1 else None)
    for _ in range(3):
        rs.render()
    rs.sync()
    t0 = time.perf_counter()
    k = 30
    for _ in range(k):
        rs.render()
    rs.sync()
    dt = (time.perf_counter() - t0) / k
    rs.stage_timing(True)
    for _ in range(5):
        rs.render()
    rs.sync()
    st, rounds = rs.stage_times_ms()
    print(f"N={n}: rank 0 renders its share in {dt*1e3:.3f} ms/frame (ideal {1.0/n:.3f} of N=1); stages/frame {{k: round(v/5, 4) for k, v in st.items()}}".replace("{k: round(v/5, 4) for k, v in st.items()}", str({k: round(v/5, 4) for k, v in st.items()})))
    rs.close()
