"""ctypes access to the CHECKERS under oracle/ (test infrastructure only; the product never imports this).

``oracle()``   -> oracle/liboracle.so, the repo's C restatement of the reference's C path.
``ref()``      -> oracle/_ref/libref_kernel.so, the reference's own kernel file compiled in place (only present where
                  it was built from /root/reference, i.e. in the build container; on a GPU box the golden fixtures minted from it stand in).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_kernel.so")


class OracleScene(C.Structure):  # rt_oracle_scene (oracle/rt_oracle.h)
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("eye", C.c_float * 4), ("eye_to_top_left", C.c_float * 4), ("left_to_right", C.c_float * 4), ("top_to_bottom", C.c_float * 4),
        ("pixel_size_inv", C.c_float),
        ("cam_start", C.c_void_p), ("cam_end", C.c_void_p), ("cam_list", C.c_void_p),
        ("sample_count", C.c_uint32),
        ("vertex", C.c_void_p),
        ("triangle_count", C.c_uint32),
        ("tri_index", C.c_void_p), ("tri_material", C.c_void_p), ("tri_uv", C.c_void_p), ("tri_normal", C.c_void_p),
        ("axes_div", C.c_int32),
        ("box_min", C.c_void_p), ("grid_start", C.c_void_p), ("grid_list", C.c_void_p),
        ("mat_size", C.c_void_p), ("mat_start", C.c_void_p), ("textures", C.c_void_p),
        ("light_count", C.c_uint32),
        ("light_type", C.c_void_p), ("light_pos", C.c_void_p), ("light_dir", C.c_void_p), ("light_col", C.c_void_p),
        ("light_radius", C.c_void_p), ("light_half_att", C.c_void_p),
        ("out_r", C.c_void_p), ("out_g", C.c_void_p), ("out_b", C.c_void_p),
    ]


class OracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("primarySamples", "primaryCandidates", "gridRays", "gridCells", "gridCandidates", "shadedHits", "texelFetches")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Float3(C.Structure):
    _fields_ = [("s", C.c_float * 4)]


_oracle = None
_ref = None


def build_oracle() -> None:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle() -> C.CDLL:
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.rt_oracle_render.argtypes = [C.POINTER(OracleScene), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(OracleStats)]
        L.rt_oracle_randf.restype = C.c_float
        L.rt_oracle_randf.argtypes = [C.POINTER(C.c_uint64), C.c_float, C.c_float]
        L.rt_oracle_sphere_point.argtypes = [C.POINTER(C.c_uint64), C.c_float, C.POINTER(C.c_float)]
        L.rt_oracle_positive_modf.restype = C.c_float
        L.rt_oracle_positive_modf.argtypes = [C.c_float]
        fp = C.POINTER(C.c_float)
        L.rt_oracle_ray_triangle.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, fp, fp]
        L.rt_oracle_box_address.argtypes = [C.c_int, C.c_void_p, fp, C.POINTER(C.c_int)]
        L.rt_oracle_bind_in_cube.argtypes = [fp, fp, fp, fp]
        L.rt_oracle_point_line_sq.restype = C.c_float
        L.rt_oracle_point_line_sq.argtypes = [fp, fp, fp]
        L.rt_oracle_grid_trace.restype = C.c_uint32
        L.rt_oracle_grid_trace.argtypes = [C.POINTER(OracleScene), fp, fp, C.c_float, C.c_float, C.c_uint32, fp, fp, fp]
        L.rt_oracle_texel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, fp, C.c_float, C.c_float, fp]
        vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
        L.rt_oracle_build_camera_list.argtypes = [u32, u32, fp, fp, fp, fp, f32, u32, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.rt_oracle_build_scene_grid.argtypes = [u32, u32, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.rt_oracle_camera_position.argtypes = [fp, fp, fp, fp, f32, fp, fp]
        L.rt_oracle_box_meets_triangle.argtypes = [fp, fp, fp, fp, fp]
        L.rt_oracle_builders_free.argtypes = [vp]
        L.rt_oracle_builders_free.restype = None
        _oracle = L
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        vp, u32, f32 = C.c_void_p, C.c_uint32, C.c_float
        L.ref_render.argtypes = [u32, u32, vp, vp, vp, vp, f32, vp, vp, vp, u32, vp, u32, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp,
                                 u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32]
        L.randF.restype = f32
        L.randF.argtypes = [C.POINTER(C.c_uint64), f32, f32]
        L.GetSpherePoint.restype = Float3
        L.GetSpherePoint.argtypes = [C.POINTER(C.c_uint64), f32]
        L.positive_modf.restype = f32
        L.positive_modf.argtypes = [f32]
        L.GetPointToLineSqLen.restype = f32
        L.GetPointToLineSqLen.argtypes = [Float3, Float3, Float3]
        L.RayIntersectsTriangle.restype = u32
        L.RayIntersectsTriangle.argtypes = [Float3, Float3, f32, f32, Float3, Float3, Float3, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
        L.GetBoxAddress.restype = type("Int3", (C.Structure,), {"_fields_": [("s", C.c_int32 * 4)]})
        L.GetBoxAddress.argtypes = [C.c_int32, vp, Float3]
        L.BindInCube.restype = u32
        L.BindInCube.argtypes = [C.POINTER(Float3), Float3, Float3, Float3]
        _ref = L
    return _ref


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def oracle_scene(sc, planes) -> OracleScene:
    o = OracleScene()
    o.width, o.height = sc.width, sc.height
    for i in range(4):
        o.eye[i] = float(sc.eye[i]); o.eye_to_top_left[i] = float(sc.eye_to_top_left[i])
        o.left_to_right[i] = float(sc.left_to_right[i]); o.top_to_bottom[i] = float(sc.top_to_bottom[i])
    o.pixel_size_inv = sc.pixel_size_inv
    o.cam_start, o.cam_end, o.cam_list = _ptr(sc.cam_start), _ptr(sc.cam_end), _ptr(sc.cam_list)
    o.sample_count = sc.sample_count
    o.vertex = _ptr(sc.vertex)
    o.triangle_count = sc.triangle_count
    o.tri_index, o.tri_material, o.tri_uv, o.tri_normal = _ptr(sc.tri_index), _ptr(sc.tri_material), _ptr(sc.tri_uv), _ptr(sc.tri_normal)
    o.axes_div = 256
    o.box_min, o.grid_start, o.grid_list = _ptr(sc.box_min), _ptr(sc.grid_start), _ptr(sc.grid_list)
    o.mat_size, o.mat_start, o.textures = _ptr(sc.mat_size), _ptr(sc.mat_start), _ptr(sc.textures)
    o.light_count = sc.light_count
    o.light_type, o.light_pos, o.light_dir, o.light_col = _ptr(sc.light_type), _ptr(sc.light_pos), _ptr(sc.light_dir), _ptr(sc.light_col)
    o.light_radius, o.light_half_att = _ptr(sc.light_radius), _ptr(sc.light_half_att)
    o.out_r, o.out_g, o.out_b = _ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2])
    return o


def oracle_render(sc, threads: int = 1, with_stats: bool = False, first_pixel: int = 0, pixel_count: int = None):
    """Runs the C restatement over the scene.  Returns [H,W] u16 R,G,B (and the work counters)."""
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    o = oracle_scene(sc, planes)
    st = OracleStats()
    n = sc.pixels - first_pixel if pixel_count is None else pixel_count
    oracle().rt_oracle_render(C.byref(o), first_pixel, n, threads, C.byref(st) if with_stats else None)
    out = [p.reshape(sc.height, sc.width) for p in planes]
    return (out, st.as_dict()) if with_stats else out


def ref_render(sc, first_pixel: int = 0, pixel_count: int = None):
    """Runs the REFERENCE kernel (compiled in place) over the scene, single thread, as raytrace.c:604-655 does."""
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    n = sc.pixels - first_pixel if pixel_count is None else pixel_count
    # the kernel takes non-const pointers and never writes its inputs
    ref().ref_render(sc.width, sc.height, _ptr(sc.eye), _ptr(sc.eye_to_top_left), _ptr(sc.left_to_right), _ptr(sc.top_to_bottom),
                     sc.pixel_size_inv, _ptr(sc.cam_start), _ptr(sc.cam_end), _ptr(sc.cam_list), sc.sample_count, _ptr(sc.vertex),
                     sc.triangle_count, _ptr(sc.tri_index), _ptr(sc.tri_material), _ptr(sc.tri_uv), _ptr(sc.tri_normal), 256,
                     _ptr(sc.box_min), _ptr(sc.grid_start), _ptr(sc.grid_list), _ptr(sc.mat_size), _ptr(sc.mat_start), _ptr(sc.textures),
                     sc.light_count, _ptr(sc.light_type), _ptr(sc.light_pos), _ptr(sc.light_dir), _ptr(sc.light_col),
                     _ptr(sc.light_radius), _ptr(sc.light_half_att), _ptr(planes[0]), _ptr(planes[1]), _ptr(planes[2]), first_pixel, n)
    return [p.reshape(sc.height, sc.width) for p in planes]


# ---- builders (oracle/rt_oracle_builders.c: independent serial restatement of trianglelist.cpp; parity unpinned) ----------

def _take_oracle(ptr, count, dtype):
    out = np.empty(int(count), dtype)
    if count:
        C.memmove(out.ctypes.data, ptr.value, int(count) * out.itemsize)
    oracle().rt_oracle_builders_free(ptr)
    return out


def _f3p(a):
    return np.ascontiguousarray(a[:3], np.float32).ctypes.data_as(C.POINTER(C.c_float))


def oracle_camera_list(sc):
    """(start[P], end[P], list[n]) as CameraTriangleList::New would build them (trianglelist.cpp:520-626)."""
    ps, pe, pl, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = oracle().rt_oracle_build_camera_list(sc.width, sc.height, _f3p(sc.eye), _f3p(sc.eye_to_top_left), _f3p(sc.left_to_right),
                                              _f3p(sc.top_to_bottom), sc.pixel_size_inv, sc.triangle_count, _ptr(sc.vertex),
                                              _ptr(sc.tri_index), C.byref(ps), C.byref(pe), C.byref(pl), C.byref(n))
    assert rc == 0
    return _take_oracle(ps, sc.pixels, np.uint32), _take_oracle(pe, sc.pixels, np.uint32), _take_oracle(pl, n.value, np.uint32)


def oracle_scene_grid(sc):
    """(box_min[257,4], start[256^3+1], list[n]) as SceneTriangleList::New would build them (trianglelist.cpp:655-737)."""
    box = np.zeros((257, 4), np.float32)
    ps, pl, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
    rc = oracle().rt_oracle_build_scene_grid(sc.vertex_count, sc.triangle_count, _ptr(sc.vertex), _ptr(sc.tri_index), _ptr(box),
                                             C.byref(ps), C.byref(pl), C.byref(n))
    assert rc == 0
    return box, _take_oracle(ps, 256 ** 3 + 1, np.uint32), _take_oracle(pl, n.value, np.uint32)
