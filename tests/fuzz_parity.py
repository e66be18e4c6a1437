"""Fuzz of the hot path against the CPU oracle on the GPU box (not collected by pytest; the oracle stays under tests/): random
materials (textures, bump maps, mirrors, transparency), random lights of every type, random soups.
usage: python tests/fuzz_parity.py [first seed] [count]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
from opencl_render_amd import raytrace as R, scene as S

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rng = np.random.Generator(np.random.PCG64(seed))
    nm = int(rng.integers(1, 6))
    mats = [dict(color=rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3)), reflection=tuple(rng.integers(0, 200, 3)),
                 transparency=tuple(rng.integers(0, 200, 3)), bump=rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3)),
                 luminance=tuple(rng.integers(0, 60, 3))) for _ in range(nm)]
    nl = int(rng.integers(0, 5))
    lights = [dict(type=int(t), pos=tuple(rng.uniform(-1, 1, 3) + [0, 0, 2]), dir=tuple(rng.uniform(-1, 1, 3)), col=tuple(rng.uniform(0, 1, 3)),
                   radius=float(rng.uniform(0, 1)), half_att=float(rng.choice([np.inf, 2.5, 0.7]))) for t in rng.integers(0, 11, nl)]
    w, h = int(rng.integers(60, 400)), int(rng.integers(40, 300))
    tris, edge = int(rng.integers(200, 20000)), float(rng.choice([0.01, 0.03, 0.08, 0.2]))
    sc = S.make_soup(w, h, tris, edge, seed=seed, samples=int(rng.integers(1, 5)), materials=mats, lights=lights, random_uv=True,
                     smooth_normals=bool(rng.integers(0, 2)))
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    got = R.render_resident(sc, 0)
    diff = sum(int((np.asarray(g).reshape(-1) != np.asarray(x).reshape(-1)).sum()) for g, x in zip(got, want))
    print(f"seed {seed}: {w}x{h}, {tris} triangles (edge {edge}), {nm} materials, {nl} lights, S={sc.sample_count}: {'OK' if diff == 0 else f'{diff} VALUES DIFFER'}", flush=True)
    bad += diff != 0
print("fuzz:", "all bit-exact" if bad == 0 else f"{bad} scenes differ")
sys.exit(1 if bad else 0)
