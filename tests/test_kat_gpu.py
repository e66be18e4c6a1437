"""Function-level known answers ON THE DEVICE (run with -m gpu): tests/golden/kat.npz was minted from the reference kernel
(tests/golden/make_golden.py); here the kernels' own building blocks (rt_devfuncs.h) run on the GPU through the test-only
entry point rtHipDeviceKat and must reproduce every answer bit for bit -- randF streams incl. seed 2^64-1,
positive_modf(-2^-30) = 1.0f, degenerate triangles, axis-parallel rays, points on split planes.  The one tolerance in the
design, (float)pow(0.5f, x) through the device's double exp2, is stated below."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from opencl_render_amd import raytrace as R

pytestmark = pytest.mark.gpu

RANDF, SPHERE, PMODF, TRI, PLINE, BOX, BIND, POW = range(8)


@pytest.fixture(scope="module")
def kat(hip_lib):
    if hip_lib.rtHipDeviceCount() < 1:
        pytest.fail("no HIP device: the device KATs cannot run (there is no CPU stand-in)")
    return np.load(os.path.join(GOLDEN, "kat.npz"))


def run(op, inp, out_stride, table=None):
    inp = np.ascontiguousarray(inp)
    n = inp.shape[0]
    in_stride = inp.nbytes // n
    out = np.zeros((n, out_stride), np.uint8)
    rc = R.lib().rtHipDeviceKat(0, op, n, inp.ctypes.data_as(C.c_void_p), in_stride, out.ctypes.data_as(C.c_void_p), out_stride,
                                None if table is None else table.ctypes.data_as(C.c_void_p))
    assert rc == 0, f"rtHipDeviceKat(op={op}) returned {rc}"
    return out


def test_device_randf_streams(kat):
    out = run(RANDF, kat["rand_seeds"].reshape(-1, 1), 192)
    draws = out[:, :64].copy().view(np.float32)
    states = out[:, 64:].copy().view(np.uint64)
    assert draws.tobytes() == kat["rand_draws"].tobytes()
    assert np.array_equal(states, kat["rand_states"])
    assert int(kat["rand_seeds"][7]) == 2 ** 64 - 1  # the all-ones seed is in the set


def test_device_sphere_point(kat):
    n = len(kat["sphere_seeds"])
    inp = np.zeros((n, 16), np.uint8)
    inp[:, :8] = kat["sphere_seeds"].reshape(-1, 1).view(np.uint8)
    inp[:, 8:12] = kat["sphere_radius"].reshape(-1, 1).view(np.uint8)
    out = run(SPHERE, inp, 24)
    assert out[:, :12].copy().view(np.float32).tobytes() == kat["sphere_out"].tobytes()
    assert np.array_equal(out[:, 16:24].copy().view(np.uint64).reshape(-1), kat["sphere_state"])


def test_device_positive_modf(kat):
    out = run(PMODF, kat["pmodf_in"].reshape(-1, 1), 4).view(np.float32).reshape(-1)
    assert out.tobytes() == kat["pmodf_out"].tobytes()
    tiny = run(PMODF, np.array([[-2.0 ** -30]], np.float32), 4).view(np.float32)
    assert tiny[0, 0] == np.float32(1.0)  # needs the 53-bit sum (raytrace_opencl.c:25-28)


def test_device_ray_triangle_and_point_line(kat):
    n = len(kat["tri_o"])
    inp = np.concatenate([kat[k] for k in ("tri_o", "tri_d", "tri_a", "tri_b", "tri_c")] + [kat["tri_tmin"].reshape(n, 1), kat["tri_tmax"].reshape(n, 1)], 1)
    assert inp.shape == (n, 17)
    out = run(TRI, inp.astype(np.float32), 20).view(np.float32)
    assert out[:, :4].tobytes() == kat["tri_res"].tobytes()
    assert (out[:, 4] == 1).all(), "the branch-free test of the wavefront kernels disagrees with tri_test"
    pl = run(PLINE, np.concatenate([kat["tri_a"], kat["tri_b"], kat["tri_o"]], 1).astype(np.float32), 4).view(np.float32).reshape(-1)
    assert pl.tobytes() == kat["pline_out"].tobytes()


def test_device_box_address_and_bind(kat):
    box = kat["box_min"]
    planes = np.ascontiguousarray(box[:, :3].T.astype(np.float32))  # one array per axis, as the scene upload lays them out
    addr = run(BOX, kat["box_pts"].astype(np.float32), 12, planes).view(np.int32)
    assert np.array_equal(addr, kat["box_addr"])
    n = len(kat["bind_o"])
    lo, hi = np.tile(box[0, :3], (n, 1)), np.tile(box[256, :3], (n, 1))
    p = run(BIND, np.concatenate([kat["bind_o"], kat["bind_d"], lo, hi], 1).astype(np.float32), 12).view(np.float32)
    assert p.tobytes() == kat["bind_res"][:, 1:].tobytes()


def test_device_half_attenuation_pow(kat):
    """(float)pow(0.5f, x): device double exp2 (OCML, < 1 ulp in double) against this container's glibc pow.  The two can
    only differ where the double result sits on a float rounding tie; north_star allows 1 ULP in fp32 shading.  Bar written
    here: at most 1 ULP anywhere, and bit-exact on at least 99.9 % of the vectors (observed so far: all)."""
    got = run(POW, kat["pow_x"].reshape(-1, 1), 4).view(np.float32).reshape(-1)
    want = kat["pow_out"]
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, f"max {ulp.max()} ULP"
    assert (ulp == 0).mean() >= 0.999, f"{(ulp != 0).sum()} of {len(ulp)} vectors differ by 1 ULP"
    assert got[kat["pow_x"] == 0][0] == 1.0


QUOTIENT = 8


def test_device_walk_quotient_short_sequence_is_the_division():
    """wf_trace_kernel's walk computes (plane - o) / d -- the correctly rounded quotient the reference's C division gives
    (raytrace_opencl.c:383-385) -- and, for waves whose operands have tame exponents, a 5-instruction sequence that is the
    compiler's division minus the scaling / fix-up instructions.  Both run here on 3 M operand triples: wherever the kernel's own
    `tame` predicate holds the two must agree bit for bit (a zero may differ in sign: the values are only ever compared), and
    the division itself must be IEEE (numpy float32).  Operands: log-uniform over the whole tame range, exponents at both ends
    of it, differences of neighbouring floats, exact zeros, significands of all ones."""
    rng = np.random.default_rng(2024)
    n = 1_000_000

    def mag(lo, hi, k):  # sign * 2^U(lo,hi) with a random significand
        e = rng.uniform(lo, hi, k)
        return (np.exp2(e) * rng.choice([-1.0, 1.0], k)).astype(np.float32)

    plane = mag(-60, 39, 3 * n)
    o = mag(-60, 39, 3 * n)
    d = mag(-40, 40, 3 * n)
    # second million: origins close to their plane (the common case on a walk), incl. neighbours and equal values
    plane[n:2 * n] = mag(-8, 8, n)
    ulps = rng.integers(-4, 5, n)
    o[n:2 * n] = (plane[n:2 * n].view(np.int32) + ulps.astype(np.int32)).view(np.float32)
    # third million: the ends of the ranges, zeros, all-ones significands
    ends = np.array([2.0 ** -60, 2.0 ** 39, -(2.0 ** -60), -(2.0 ** 39), 0.0, np.float32(2.0 ** 39) * np.float32(1 - 2.0 ** -24)], np.float32)
    dends = np.array([2.0 ** -40, 2.0 ** 40, -(2.0 ** -40), -(2.0 ** 40), np.float32(2.0 ** -39) * np.float32(1 - 2.0 ** -24), 1.0, 3.0], np.float32)
    plane[2 * n:] = rng.choice(ends, n)
    o[2 * n:] = np.where(rng.random(n) < 0.5, rng.choice(ends, n), o[2 * n:])
    d[2 * n:] = np.where(rng.random(n) < 0.5, rng.choice(dends, n), (d[2 * n:].view(np.uint32) | np.uint32(0x7fffff)).view(np.float32))
    inp = np.stack([plane, o, d], axis=1).astype(np.float32)
    out = run(QUOTIENT, inp, 12)
    exact = out[:, 0:4].copy().view(np.float32).reshape(-1)
    short = out[:, 4:8].copy().view(np.float32).reshape(-1)
    tame = out[:, 8:12].copy().view(np.uint32).reshape(-1)
    assert tame.mean() > 0.99  # the operands were drawn from the tame ranges (all-ones significands at the very end fall out)
    with np.errstate(all="ignore"):
        want = ((plane - o).astype(np.float32) / d).astype(np.float32)
    assert exact.tobytes() == want.tobytes(), "the device division is not the IEEE quotient"
    t = tame == 1
    same = (exact.view(np.uint32) == short.view(np.uint32)) | ((exact == 0) & (short == 0))
    bad = np.flatnonzero(t & ~same)
    assert bad.size == 0, f"short sequence differs on {bad.size} tame triples, first {inp[bad[0]]}: {exact[bad[0]]!r} vs {short[bad[0]]!r}"
    # and the predicate refuses what the short sequence cannot do
    wild = np.array([[1.0, 0.5, 0.0], [1.0, 0.5, 1e-30], [1e30, 0.5, 1.0], [1.0, 1e-25, 1.0], [np.inf, 0.0, 1.0], [1.0, np.nan, 1.0],
                     [1.0, 0.5, np.inf], [1.0, 0.5, 2.0 ** 41]], np.float32)
    assert not run(QUOTIENT, wild, 12)[:, 8:12].copy().view(np.uint32).any()
