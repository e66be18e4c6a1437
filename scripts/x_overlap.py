"""Experiment: two frames in flight.  Two instances of the scene on one GPU, each on its own stream, frames issued alternately
(frame i+1's kernels fill what frame i's draining kernels leave of the chip), against the same frames one after the other on one stream.
usage: python scripts/x_overlap.py [workload] [--samples S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from opencl_render_amd import raytrace as R
args = sys.argv[1:]
samples = 1
if "--samples" in args:
    i = args.index("--samples")
    samples = int(args[i + 1])
    del args[i:i + 2]
wl = args[0] if args else "lambert_1m"
sc = bench.make_scene(wl, samples)
dev = torch.device("cuda", 0)
inst = [R.ResidentScene(sc, 0, None) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
k = 40 if sc.pixels * samples < 20_000_000 else 10


def run(n_inst, frames):
    for i in range(frames):
        inst[i % n_inst].render(streams[i % n_inst].cuda_stream)


for n_inst in (1, 2, 1, 2):
    run(n_inst, 6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n_inst, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    bad = [rs.finish() for rs in inst]
    print(f"{wl} S={samples}: {n_inst} frame(s) in flight: {dt * 1e3:.3f} ms/frame (frames redone after the clock stopped: {bad})", flush=True)
a = [x.copy() for x in inst[0].readback()]
b = inst[1].readback()
print("both instances hold the same planes:", all((x == y).all() for x, y in zip(a, b)))
for rs in inst:
    rs.close()
