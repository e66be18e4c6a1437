"""CPU tests: the oracle (C restatement) against the golden fixtures minted from the reference kernel, against the
function-level known answers, and -- where oracle/_ref exists (the build container) -- against the reference kernel
itself on freshly generated scenes."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import scenarios
from conftest import GOLDEN, golden_names, load_golden_scene


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden_planes(name):
    sc, want = load_golden_scene(name)
    got = O.oracle_render(sc, threads=1)
    for ch, g, w in zip("RGB", got, want):
        assert np.array_equal(g, w), f"{name}: plane {ch} differs in {(g != w).sum()} pixels"


def test_oracle_threads_do_not_change_planes():
    sc, want = load_golden_scene("sparse_many_samples")
    got = O.oracle_render(sc, threads=4)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_oracle_pixel_ranges_compose():
    sc, want = load_golden_scene("lambert_distant")
    half = sc.pixels // 2
    a = O.oracle_render(sc, first_pixel=0, pixel_count=half)
    b = O.oracle_render(sc, first_pixel=half, pixel_count=sc.pixels - half)
    for x, y, w in zip(a, b, want):
        assert np.array_equal(x + y, w)


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "kat.npz"))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def test_kat_randf(kat):
    L = O.oracle()
    for i, seed in enumerate(kat["rand_seeds"]):
        st = C.c_uint64(int(seed))
        for j in range(16):
            lo, hi = ((0.0, 1.0), (-1.0, 1.0))[j & 1]
            v = np.float32(L.rt_oracle_randf(C.byref(st), lo, hi))
            assert v.tobytes() == kat["rand_draws"][i, j].tobytes()
            assert st.value == int(kat["rand_states"][i, j])


def test_kat_sphere_point(kat):
    L = O.oracle()
    for i in range(len(kat["sphere_seeds"])):
        st = C.c_uint64(int(kat["sphere_seeds"][i]))
        out = np.zeros(3, np.float32)
        L.rt_oracle_sphere_point(C.byref(st), float(kat["sphere_radius"][i]), _fp(out))
        assert out.tobytes() == kat["sphere_out"][i].tobytes()
        assert st.value == int(kat["sphere_state"][i])


def test_kat_positive_modf(kat):
    L = O.oracle()
    got = np.array([L.rt_oracle_positive_modf(float(v)) for v in kat["pmodf_in"]], np.float32)
    assert got.tobytes() == kat["pmodf_out"].tobytes()
    assert np.float32(L.rt_oracle_positive_modf(-2.0 ** -30)) == np.float32(1.0)  # the 53-bit sum matters


def test_kat_ray_triangle_and_point_line(kat):
    L = O.oracle()
    n = len(kat["tri_o"])
    for i in range(n):
        t, l1, l2 = C.c_float(0), C.c_float(0), C.c_float(0)
        arrs = [np.ascontiguousarray(kat[k][i]) for k in ("tri_o", "tri_d", "tri_a", "tri_b", "tri_c")]
        hit = L.rt_oracle_ray_triangle(_fp(arrs[0]), _fp(arrs[1]), float(kat["tri_tmin"][i]), float(kat["tri_tmax"][i]),
                                       _fp(arrs[2]), _fp(arrs[3]), _fp(arrs[4]), C.byref(t), C.byref(l1), C.byref(l2))
        got = np.array([hit, t.value, l1.value, l2.value], np.float32)
        assert got.tobytes() == kat["tri_res"][i].tobytes(), i
        pl = np.float32(L.rt_oracle_point_line_sq(_fp(arrs[2]), _fp(arrs[3]), _fp(arrs[0])))
        assert pl.tobytes() == kat["pline_out"][i].tobytes(), i


def test_kat_box_address_and_bind(kat):
    L = O.oracle()
    box = np.ascontiguousarray(kat["box_min"])
    for i, p in enumerate(kat["box_pts"]):
        out = (C.c_int * 3)()
        pp = np.ascontiguousarray(p)
        L.rt_oracle_box_address(256, box.ctypes.data_as(C.c_void_p), _fp(pp), out)
        assert list(out) == list(kat["box_addr"][i]), i
    lo, hi = np.ascontiguousarray(box[0, :3]), np.ascontiguousarray(box[256, :3])
    for i in range(len(kat["bind_o"])):
        p = np.ascontiguousarray(kat["bind_o"][i]).copy()
        d = np.ascontiguousarray(kat["bind_d"][i])
        ok = L.rt_oracle_bind_in_cube(_fp(p), _fp(d), _fp(lo), _fp(hi))
        got = np.array([ok, *p], np.float32)
        assert got.tobytes() == kat["bind_res"][i].tobytes(), i


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference; build container only)")
@pytest.mark.parametrize("seed", [101, 202, 303])
def test_oracle_matches_reference_kernel_on_fresh_scenes(seed):
    """Fresh seeds, not the committed fixtures: restatement == reference kernel, plane for plane."""
    from opencl_render_amd import raytrace as R, scene as S
    rng = np.random.Generator(np.random.PCG64(seed))
    mats = [dict(color=tuple(rng.integers(0, 256, 3)), reflection=tuple(rng.integers(0, 200, 3)),
                 transparency=tuple(rng.integers(0, 200, 3)), bump=rng.integers(0, 256, (5, 5, 3)),
                 luminance=tuple(rng.integers(0, 60, 3))) for _ in range(3)]
    lights = [dict(type=int(rng.integers(0, 10)), pos=tuple(rng.uniform(-1, 1, 3)), dir=tuple(rng.uniform(-1, 1, 3)),
                   col=tuple(rng.uniform(0, 1, 3)), radius=float(rng.uniform(0, 1)), half_att=float(rng.choice([np.inf, 1.5])))
              for _ in range(3)]
    sc = S.make_soup(40, 32, 500, 0.3, seed=seed, samples=2, materials=mats, lights=lights, random_uv=True, smooth_normals=True)
    R.build_lists(sc)
    a = O.oracle_render(sc)
    b = O.ref_render(sc)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
