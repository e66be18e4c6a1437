"""Times one rank's share of the frame on one GPU: what a rank of an N-GPU run does per frame, without the gather.
usage: python scripts/rank_share.py [workload] [--samples S] [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opencl_render_amd import raytrace as R
args = sys.argv[1:]
samples = 1
if "--samples" in args:
    i = args.index("--samples")
    samples = int(args[i + 1])
    del args[i:i + 2]
wl = args[0] if args else "lambert_1m"
ns = [int(a) for a in args[1:]] or [1, 2, 4, 8]
sc = bench.make_scene(wl, samples)
base = None
for n in ns:
    rs = R.ResidentScene(sc, 0, R.tiles_of_rank(sc.width, sc.height, 0, n) if n > 1 else None)
    # (the chip needs ~20 ms of continuous work to reach its steady clocks from idle -- scripts/x_ramp.py; ranks of a real run render
    # frame after frame, so the share is timed on a busy chip)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        for _ in range(5):
            rs.render()
        rs.sync()
    t0 = time.perf_counter()
    k = 30 if sc.pixels * samples < 20_000_000 else 8
    for _ in range(k):
        rs.render()
    rs.sync()
    dt = (time.perf_counter() - t0) / k
    rs.stage_timing(True)
    for _ in range(5):
        rs.render()
    rs.sync()
    st, rounds = rs.stage_times_ms()
    if n == 1:
        base = dt
    bound = f", speed-up bound before the gather {base / dt:.2f}x" if base else ""
    print(f"{wl} S={samples} N={n}: rank 0 renders its share in {dt*1e3:.3f} ms/frame (ideal {1.0/n:.3f} of N=1{bound}); "
          f"stages/frame {str({k_: round(v/5, 4) for k_, v in st.items()})}", flush=True)
    rs.close()
