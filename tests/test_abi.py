"""CPU tests of the C-ABI library: it loads, exports every symbol include/raytrace_hip.h declares, its host-side
helpers compute the reference's values, and the computing entry points FAIL LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from opencl_render_amd import raytrace as R


def declared_functions():
    text = open(os.path.join(ROOT, "include", "raytrace_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    skip = {"__attribute__", "aligned", "defined", "RaytraceAll" if False else ""}
    return sorted({n for n in names if n not in skip and not n.startswith("__")})


def test_header_and_binding_agree_on_the_symbol_list():
    assert set(declared_functions()) == set(R.DROPIN_SYMBOLS + R.RESIDENT_SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_functions():
        assert hasattr(hip_lib, name), f"libraytrace_hip.so does not export {name}"


def test_no_product_file_touches_the_oracle():
    """The product must never route through oracle/ (only tests/, smoke() and bench.py's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "opencl_render_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "rt_oracle" not in text and "oracle_lib" not in text and "liboracle" not in text, f


def f3(v):
    out = R.Float3()
    for i in range(3):
        out.s[i] = float(v[i])
    return out


def test_math_helpers_match_reference_known_answers(hip_lib):
    kat = np.load(os.path.join(GOLDEN, "kat.npz"))
    n = len(kat["tri_o"])
    for i in range(n):
        t, l1, l2 = C.c_float(0), C.c_float(0), C.c_float(0)
        hit = hip_lib.RayIntersectsTriangle(f3(kat["tri_o"][i]), f3(kat["tri_d"][i]), float(kat["tri_tmin"][i]), float(kat["tri_tmax"][i]),
                                            f3(kat["tri_a"][i]), f3(kat["tri_b"][i]), f3(kat["tri_c"][i]), C.byref(t), C.byref(l1), C.byref(l2))
        got = np.array([hit, t.value, l1.value, l2.value], np.float32)
        assert got.tobytes() == kat["tri_res"][i].tobytes(), i
        pl = np.float32(hip_lib.GetPointToLineSqLen(f3(kat["tri_a"][i]), f3(kat["tri_b"][i]), f3(kat["tri_o"][i])))
        assert pl.tobytes() == kat["pline_out"][i].tobytes(), i
    box = np.ascontiguousarray(kat["box_min"])
    for i, p in enumerate(kat["box_pts"]):
        a = hip_lib.GetBoxAddress(256, box.ctypes.data_as(C.c_void_p), f3(p))
        assert [a.s[0], a.s[1], a.s[2]] == list(kat["box_addr"][i]), i


def test_small_vector_helpers(hip_lib):
    a, b = (1.0, 2.0, 3.0), (-4.0, 0.5, 2.0)
    assert hip_lib.dot(f3(a), f3(b)) == np.float32(1 * -4 + 2 * 0.5) + np.float32(3 * 2)
    c = hip_lib.cross(f3(a), f3(b))
    assert [c.s[0], c.s[1], c.s[2]] == [2 * 2 - 3 * 0.5, 3 * -4 - 1 * 2, 1 * 0.5 - 2 * -4]
    v = hip_lib.vector(f3(a), f3(b))
    assert [v.s[0], v.s[1], v.s[2]] == [-5.0, -1.5, -1.0]
    n = hip_lib.normalize(f3((3.0, 0.0, 4.0)))
    assert [n.s[0], n.s[1], n.s[2]] == [np.float32(3.0) / np.float32(5.0), 0.0, np.float32(4.0) / np.float32(5.0)]
    assert hip_lib.bindf(5.0, 0.0, 1.0) == 1.0 and hip_lib.bindf(-5.0, 0.0, 1.0) == 0.0 and hip_lib.bindf(0.25, 0.0, 1.0) == 0.25


def test_computation_type_table(hip_lib, gpu_count):
    names = R.computation_type_names()
    assert names[0] == "Local CPU single thread"  # reference raytrace.c:138
    assert hip_lib.GetIsComputationTypeUpdated() == 1
    expect = 1 + gpu_count + (1 if gpu_count > 1 else 0)
    assert hip_lib.GetComputationTypeCount() == expect
    buf = C.create_string_buffer(8)
    assert hip_lib.GetComputationTypeName(0, 4, buf) == 0  # does not fit -> CL_FALSE (raytrace.c:139,152)
    hip_lib.ResetComputationType()
    assert hip_lib.GetIsComputationTypeUpdated() == 0 and hip_lib.GetComputationTypeCount() == 1
    hip_lib.InitOpenCL()


def test_progress_and_time_accessors(hip_lib):
    hip_lib.SetProgress(0.25)
    assert hip_lib.GetProgress() == 0.25
    hip_lib.ResetTime()
    assert hip_lib.GetStartTime() == 0 and hip_lib.GetEndTime() == 0


def test_type_zero_is_refused_loudly(capfd):
    """computationType 0 is the reference's own CPU loop; this library must not quietly compute on the CPU."""
    from conftest import load_golden_scene
    sc, _ = load_golden_scene("no_material")
    ok, r, g, b = R.raytrace_all(0, sc)
    assert not ok and not r.any()
    assert "computationType 0" in R.last_error()
    assert "libraytrace_hip" in capfd.readouterr().err


def test_without_a_gpu_everything_that_computes_fails(gpu_count):
    if gpu_count > 0:
        pytest.skip("a HIP device is present")
    from conftest import load_golden_scene
    sc, _ = load_golden_scene("no_material")
    ok, r, g, b = R.raytrace_all(1, sc)
    assert not ok
    with pytest.raises(RuntimeError, match="no HIP device"):
        R.ResidentScene(sc, 0)


def test_cache_hash_sees_every_tail_byte(hip_lib):
    """The scene cache decides 'unchanged' from content hashes alone: an edit confined to the last bytes of an array must
    change the hash.  The cases are the ones an ORed tail word missed (light 2's bits a subset of light 0's)."""
    def h(a):
        a = np.ascontiguousarray(a)
        return hip_lib.rtHipTestHashBytes(a.ctypes.data_as(C.c_void_p), a.nbytes)
    assert h(np.array([1, 2, 0.5], np.float32)) != h(np.array([1, 2, 1.0], np.float32))       # lightRadius, 12-byte tail
    assert h(np.array([3, 0, 1], np.int32)) != h(np.array([3, 0, 2], np.int32))                   # lightType
    assert h(np.array([7, 7, 7, 7, 7, 7, 3], np.int32)) != h(np.array([7, 7, 7, 7, 7, 7, 1], np.int32))  # triMaterial, T % 8 == 7
    rng = np.random.default_rng(5)
    for n in list(range(1, 100)) + [255, 1000, 4097]:
        base = rng.integers(0, 256, n, dtype=np.uint8)
        seen = {h(base)}
        for i in range(max(0, n - 40), n):  # flip every bit pattern position in the last 40 bytes, one byte at a time
            for bit in (1, 0x80, 0xff):
                e = base.copy()
                e[i] ^= bit
                v = h(e)
                assert v not in seen, (n, i, bit)
        assert h(base[:-1]) != h(base) if n > 1 else True
