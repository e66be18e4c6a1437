#!/usr/bin/env python3
"""Mints the golden fixtures under tests/golden/ from the REFERENCE's own kernel.

Run in the build container only (it needs oracle/_ref/libref_kernel.so, i.e. /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Provenance of every number written here: oracle/_ref/libref_kernel.so is /root/reference/source/opencl/
raytrace_opencl.c compiled in place by oracle/Makefile (plus the 3 OpenCL built-ins and the pixel/sample loop in
oracle/ref_glue.c); the reference ships no golden vectors of its own.  Fixtures are data only:
scene inputs in the ABI layouts and the expected u16 planes / function results.

  scene_<name>.npz   inputs (grid start stored sparsely as non-empty cell ids + counts) and planes R,G,B
  kat.npz            function-level known answers (randF, GetSpherePoint, positive_modf, RayIntersectsTriangle,
                     GetPointToLineSqLen, GetBoxAddress, BindInCube)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
import scenarios  # noqa: E402
from opencl_render_amd import raytrace as R  # noqa: E402

SCENE_FIELDS = ["eye", "eye_to_top_left", "left_to_right", "top_to_bottom", "vertex", "tri_index", "tri_material", "tri_uv",
                "tri_normal", "mat_size", "mat_start", "textures", "light_type", "light_pos", "light_dir", "light_col",
                "light_radius", "light_half_att", "cam_start", "cam_end", "cam_list", "box_min", "grid_list"]


def save_scene(sc, planes, path):
    counts = np.diff(sc.grid_start.astype(np.int64))
    cells = np.nonzero(counts)[0].astype(np.uint32)
    data = {k: getattr(sc, k) for k in SCENE_FIELDS}
    data.update(grid_cells=cells, grid_counts=counts[cells].astype(np.uint32),
                dims=np.array([sc.width, sc.height, sc.sample_count], np.uint32),
                pixel_size_inv=np.float32(sc.pixel_size_inv), out_r=planes[0], out_g=planes[1], out_b=planes[2])
    np.savez_compressed(path, **data)


def f3(v):
    out = O.Float3()
    for i in range(3):
        out.s[i] = float(v[i])
    return out


def make_kat(path):
    ref = O.ref()
    rng = np.random.Generator(np.random.PCG64(2024))
    kat = {}
    # randF streams (raytrace_opencl.c:12-23)
    seeds = np.array([0, 1, 2, 12345, 2**32 - 1, 2**32, 2**63, 2**64 - 1, 987654321987654321], np.uint64)
    draws = np.zeros((len(seeds), 16), np.float32)
    states = np.zeros((len(seeds), 16), np.uint64)
    for i, s in enumerate(seeds):
        st = C.c_uint64(int(s))
        for j in range(16):
            lo, hi = ((0.0, 1.0), (-1.0, 1.0))[j & 1]
            draws[i, j] = ref.randF(C.byref(st), lo, hi)
            states[i, j] = st.value
    kat.update(rand_seeds=seeds, rand_draws=draws, rand_states=states)
    # GetSpherePoint (:30-45)
    sp_seeds = rng.integers(0, 2**63, 64).astype(np.uint64)
    sp_radius = rng.uniform(0.0, 3.0, 64).astype(np.float32)
    sp_out = np.zeros((64, 3), np.float32)
    sp_state = np.zeros(64, np.uint64)
    for i in range(64):
        st = C.c_uint64(int(sp_seeds[i]))
        p = ref.GetSpherePoint(C.byref(st), float(sp_radius[i]))
        sp_out[i] = p.s[0], p.s[1], p.s[2]
        sp_state[i] = st.value
    kat.update(sphere_seeds=sp_seeds, sphere_radius=sp_radius, sphere_out=sp_out, sphere_state=sp_state)
    # positive_modf (:25-28), including tiny negatives that round to 1.0f
    pm_in = np.concatenate([rng.uniform(-5, 5, 200), [-0.0, 0.0, 1.0, -1.0, -2.0 ** -30, 2.0 ** -30, -1e-8, 123456.75, -123456.75, 0.99999994]]).astype(np.float32)
    pm_out = np.array([ref.positive_modf(float(v)) for v in pm_in], np.float32)
    kat.update(pmodf_in=pm_in, pmodf_out=pm_out)
    # RayIntersectsTriangle (:124-172): random rays against random triangles, plus degenerate triangles
    n = 400
    ro = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    rd = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    ta = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    tb = ta + rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    tc = ta + rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    tb[::17] = ta[::17]           # zero-area
    rd[::23, 2] = 0               # axis-parallel rays
    tmin = np.where(np.arange(n) % 3 == 0, np.float32(0.2), np.float32(0)).astype(np.float32)
    tmax = np.where(np.arange(n) % 5 == 0, np.float32(1.5), np.float32(np.inf)).astype(np.float32)
    res = np.zeros((n, 4), np.float32)  # hit, t, abL, acL (abL/acL = 0 when not written)
    for i in range(n):
        t, l1, l2 = C.c_float(0), C.c_float(0), C.c_float(0)
        hit = ref.RayIntersectsTriangle(f3(ro[i]), f3(rd[i]), float(tmin[i]), float(tmax[i]), f3(ta[i]), f3(tb[i]), f3(tc[i]),
                                        C.byref(t), C.byref(l1), C.byref(l2))
        res[i] = hit, t.value, l1.value, l2.value
    kat.update(tri_o=ro, tri_d=rd, tri_a=ta, tri_b=tb, tri_c=tc, tri_tmin=tmin, tri_tmax=tmax, tri_res=res)
    # GetPointToLineSqLen (:83-101)
    pl = np.array([ref.GetPointToLineSqLen(f3(ta[i]), f3(tb[i]), f3(ro[i])) for i in range(n)], np.float32)
    kat.update(pline_out=pl)
    # GetBoxAddress (:174-193) and BindInCube (:265-322) on a real grid
    sc = scenarios.lambert_distant()
    R.build_scene_grid(sc)
    pts = np.concatenate([rng.uniform(-3, 3, (300, 3)) * [1, 1, 0] + [0, 0, 3], sc.box_min[[0, 1, 128, 255, 256], :3],
                          sc.vertex[:50, :3]]).astype(np.float32)
    addr = np.zeros((len(pts), 3), np.int32)
    for i, p in enumerate(pts):
        a = ref.GetBoxAddress(256, sc.box_min.ctypes.data_as(C.c_void_p), f3(p))
        addr[i] = a.s[0], a.s[1], a.s[2]
    kat.update(box_min=sc.box_min, box_pts=pts, box_addr=addr)
    bo = (rng.uniform(-4, 4, (300, 3)) + [0, 0, 3]).astype(np.float32)
    bd = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    bd[::13, 0] = 0
    bres = np.zeros((300, 4), np.float32)
    for i in range(300):
        p = f3(bo[i])
        ok = ref.BindInCube(C.byref(p), f3(bd[i]), f3(sc.box_min[0]), f3(sc.box_min[256]))
        bres[i] = ok, p.s[0], p.s[1], p.s[2]
    kat.update(bind_o=bo, bind_d=bd, bind_res=bres)
    # pow(0.5f, maxLen/halfAtt) cast to float (:631).  Not a reference symbol: the call goes to the C library, so the known
    # answers are THIS container's glibc pow -- the library the reference's C path calls here (parity-unpinned by
    # the reference itself).  x = the float quotient the kernel forms first.
    libm = C.CDLL("libm.so.6")
    libm.pow.restype = C.c_double
    libm.pow.argtypes = [C.c_double, C.c_double]
    px = np.concatenate([rng.uniform(0, 40, 600), rng.uniform(0, 1e-3, 100), [0.0, 1.0, 2.0, 126.0, 127.0, 149.0, 150.0, 200.0, np.inf]]).astype(np.float32)
    pw = np.array([libm.pow(0.5, float(v)) for v in px], np.float64).astype(np.float32)
    kat.update(pow_x=px, pow_out=pw)
    np.savez_compressed(path, **kat)


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/libref_kernel.so missing: run `make -C oracle` where /root/reference exists")
    only = set(sys.argv[1:])  # optional: names of the fixtures to (re)write, "kat" for kat.npz; default = everything
    for f in scenarios.ALL:
        if only and f.__name__ not in only:
            continue
        sc = f()
        R.build_lists(sc)
        planes = O.ref_render(sc)
        path = os.path.join(HERE, f"scene_{sc.name}.npz")
        save_scene(sc, planes, path)
        print(f"{sc.name}: {os.path.getsize(path) / 1024:.0f} KiB, lit pixels {(planes[0] > 0).mean():.2f}")
    if not only or "kat" in only:
        make_kat(os.path.join(HERE, "kat.npz"))
        print("kat.npz written")


if __name__ == "__main__":
    main()
