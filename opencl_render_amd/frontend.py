"""Headless scene front-end and output sinks: ctypes mirror of section (3) of include/raytrace_hip.h (rt_frontend.cpp).

Counterparts of the SDK-free parts of the reference's scene extraction (``source/render.cpp``): ``SetCamera`` (:461-491),
the SoA contract of ``AddPolygonsRecursive`` (:707-1003: quads -> two triangles, corner normals or camera-facing face
normals, UVs or the fallback triple, lights), the material channel table rules (:1136-1309) and the output path
(``u16 / 256`` -> 8 bit, :1372-1386; the BMP layout of ``writebmp3s``, ``source/util/writebmp.cpp:124-177``).
Everything computes in the C++ library; this file only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import raytrace as R
from .scene import CH_COUNT, Scene


class _Mesh(C.Structure):  # rtHipMesh
    _fields_ = [("pointCount", C.c_uint32), ("points", C.c_void_p), ("polygonCount", C.c_uint32), ("polygons", C.c_void_p),
                ("cornerNormals", C.c_void_p), ("cornerUv", C.c_void_p), ("polygonMaterial", C.c_void_p)]


class _ObjMaterial(C.Structure):  # rtHipObjMaterial
    _fields_ = [("name", C.c_char * 64), ("kd", C.c_float * 3), ("ke", C.c_float * 3), ("hasKe", C.c_int32), ("dissolve", C.c_float),
                ("reflect", C.c_float), ("hasReflect", C.c_int32), ("map", (C.c_char * 256) * 5)]


class _ObjData(C.Structure):  # rtHipObjData
    _fields_ = [("pointCount", C.c_uint32), ("points", C.c_void_p), ("polygonCount", C.c_uint32), ("polygons", C.c_void_p),
                ("cornerNormals", C.c_void_p), ("cornerUv", C.c_void_p), ("polygonMaterial", C.c_void_p),
                ("materialCount", C.c_uint32), ("materials", C.POINTER(_ObjMaterial))]


class _Channel(C.Structure):  # rtHipChannelSpec
    _fields_ = [("enabled", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32), ("pixels", C.c_void_p)]


class _Material(C.Structure):  # rtHipMaterialSpec
    _fields_ = [("channel", _Channel * 5), ("color", C.c_float * 3), ("brightness", C.c_float)]


_bound = False


def _lib():
    global _bound
    L = R.lib()
    if not _bound:
        vp, u32, f32, fp = C.c_void_p, C.c_uint32, C.c_float, C.POINTER(C.c_float)
        L.rtHipSetCamera.restype = None
        L.rtHipSetCamera.argtypes = [vp, vp, vp, fp, fp, fp, fp, f32, u32, u32]
        L.rtHipMeshCount.argtypes = [C.POINTER(_Mesh), u32, C.POINTER(u32), C.POINTER(u32)]
        L.rtHipMeshFill.argtypes = [C.POINTER(_Mesh), u32, fp, vp, vp, vp, vp, vp]
        L.rtHipLightFill.restype = None
        L.rtHipLightFill.argtypes = [u32, C.c_int32, fp, fp, fp, f32, vp, vp, vp, vp, vp, vp]
        L.rtHipBakeMaterials.argtypes = [C.POINTER(_Material), u32, vp, vp, vp, u32, C.POINTER(u32)]
        L.rtHipPlanesToRgb8.restype = None
        L.rtHipPlanesToRgb8.argtypes = [u32, u32, vp, vp, vp, vp, C.c_int]
        L.rtHipWriteBmp.argtypes = [C.c_char_p, u32, u32, vp, vp, vp, C.c_int]
        L.rtHipWritePpm.argtypes = [C.c_char_p, u32, u32, vp, vp, vp]
        L.rtHipObjRead.argtypes = [C.c_char_p, C.POINTER(_ObjData)]
        L.rtHipObjFree.restype = None
        L.rtHipObjFree.argtypes = [C.POINTER(_ObjData)]
        L.rtHipImageRead.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(vp)]
        L.rtHipProjectUv.argtypes = [C.c_int, fp, fp, f32, f32, f32, f32, C.c_int, fp]
        _bound = True
    return L


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, np.float32)[:3]).ctypes.data_as(C.POINTER(C.c_float))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_camera(position, look_at, up, fov: float, width: int, height: int):
    """SetCamera (render.cpp:461-491).  Returns (eye_to_top_left[4], left_to_right[4], top_to_bottom[4], pixel_size_inv)."""
    tl, lr, tb = (np.zeros(4, np.float32) for _ in range(3))
    inv = C.c_float()
    _lib().rtHipSetCamera(_p(tl), _p(lr), _p(tb), C.byref(inv), _f3(position), _f3(look_at), _f3(up), float(fov), width, height)
    return tl, lr, tb, float(inv.value)


@dataclass
class Mesh:
    """One polygon object in world space.  polygons: [n,4] ints a,b,c,d (c == d marks a triangle)."""
    points: np.ndarray                       # [p,3] float
    polygons: np.ndarray                     # [n,4] int
    corner_normals: Optional[np.ndarray] = None  # [n,4,3] float, any length
    corner_uv: Optional[np.ndarray] = None       # [n,4,2] float
    polygon_material: Optional[np.ndarray] = None  # [n] int, -1 = none


def _pad(a, width, dtype):
    out = np.zeros(a.shape[:-1] + (width,), dtype)
    out[..., :a.shape[-1]] = a
    return np.ascontiguousarray(out)


def mesh_arrays(meshes: Sequence[Mesh], camera_eye):
    """Count + fill (render.cpp:676-963).  Returns vertex[V,4], tri_index[T,4], tri_material[T], tri_uv[3T,2], tri_normal[3T,4]."""
    L = _lib()
    keep = []
    arr = (_Mesh * len(meshes))()
    for i, m in enumerate(meshes):
        pts = _pad(np.asarray(m.points, np.float32), 4, np.float32)
        pol = np.ascontiguousarray(np.asarray(m.polygons, np.int32).reshape(-1, 4))
        nrm = None if m.corner_normals is None else _pad(np.asarray(m.corner_normals, np.float32).reshape(len(pol), 4, 3), 4, np.float32)
        uv = None if m.corner_uv is None else np.ascontiguousarray(np.asarray(m.corner_uv, np.float32).reshape(len(pol), 4, 2))
        mat = None if m.polygon_material is None else np.ascontiguousarray(np.asarray(m.polygon_material, np.int32))
        keep.append((pts, pol, nrm, uv, mat))
        arr[i].pointCount, arr[i].points = len(pts), pts.ctypes.data
        arr[i].polygonCount, arr[i].polygons = len(pol), pol.ctypes.data
        arr[i].cornerNormals = None if nrm is None else nrm.ctypes.data
        arr[i].cornerUv = None if uv is None else uv.ctypes.data
        arr[i].polygonMaterial = None if mat is None else mat.ctypes.data
    nv, nt = C.c_uint32(), C.c_uint32()
    rc = L.rtHipMeshCount(arr, len(meshes), C.byref(nv), C.byref(nt))
    if rc != 0:
        raise ValueError(f"rtHipMeshCount failed ({rc})")
    vertex = np.zeros((nv.value, 4), np.float32)
    tri_index = np.zeros((nt.value, 4), np.int32)
    tri_material = np.zeros(nt.value, np.int32)
    tri_uv = np.zeros((3 * nt.value, 2), np.float32)
    tri_normal = np.zeros((3 * nt.value, 4), np.float32)
    rc = L.rtHipMeshFill(arr, len(meshes), _f3(camera_eye), _p(vertex), _p(tri_index), _p(tri_material), _p(tri_uv), _p(tri_normal))
    if rc != 0:
        raise ValueError(f"rtHipMeshFill failed ({rc})")
    return vertex, tri_index, tri_material, tri_uv, tri_normal


def light_arrays(lights: Sequence[dict]):
    """lights: dicts with type, pos, dir, col, brightness (render.cpp:965-993)."""
    n = len(lights)
    ltype = np.zeros(n, np.int32)
    pos, direction, col = (np.zeros((n, 4), np.float32) for _ in range(3))
    radius, half = np.zeros(n, np.float32), np.zeros(n, np.float32)
    for i, l in enumerate(lights):
        _lib().rtHipLightFill(i, int(l["type"]), _f3(l.get("pos", (0, 0, 0))), _f3(l.get("dir", (0, 0, 1))), _f3(l.get("col", (1, 1, 1))),
                              float(l.get("brightness", 1.0)), _p(ltype), _p(pos), _p(direction), _p(col), _p(radius), _p(half))
    return ltype, pos, direction, col, radius, half


CHANNEL_KEYS = ("color", "reflection", "transparency", "bump", "luminance")


def bake_materials(materials: Sequence[dict]):
    """materials: dicts; per channel key either absent/None (off), True (switched on without an image) or an [h,w,3] uint8
    image; plus optional "rgb" (material colour, default 1,1,1) and "brightness" (default 1).  Returns
    (mat_size[5M,2] u32, mat_start[5M+1] i32, textures[texels,4] u8) by the rules of render.cpp:1136-1309."""
    L = _lib()
    n = len(materials)
    specs = (_Material * n)()
    keep = []
    for i, m in enumerate(materials):
        for c, key in enumerate(CHANNEL_KEYS):
            v = m.get(key)
            ch = specs[i].channel[c]
            if v is None or v is False:
                ch.enabled, ch.width, ch.height, ch.pixels = 0, 0, 0, None
            elif v is True:
                ch.enabled, ch.width, ch.height, ch.pixels = 1, 0, 0, None
            else:
                img = np.asarray(v, np.uint8)
                px = _pad(img.reshape(-1, 3), 4, np.uint8)
                keep.append(px)
                ch.enabled, ch.width, ch.height, ch.pixels = 1, img.shape[1], img.shape[0], px.ctypes.data
        rgb = m.get("rgb", (1.0, 1.0, 1.0))
        for k in range(3):
            specs[i].color[k] = float(rgb[k])
        specs[i].brightness = float(m.get("brightness", 1.0))
    size = np.zeros((CH_COUNT * n, 2), np.uint32)
    start = np.zeros(CH_COUNT * n + 1, np.int32)
    used = C.c_uint32()
    rc = L.rtHipBakeMaterials(specs, n, _p(size), _p(start), None, 0, C.byref(used))
    if rc != 0:
        raise ValueError(f"rtHipBakeMaterials (sizing) failed ({rc})")
    tex = np.zeros((used.value, 4), np.uint8)
    rc = L.rtHipBakeMaterials(specs, n, _p(size), _p(start), _p(tex), used.value, C.byref(used))
    if rc != 0:
        raise ValueError(f"rtHipBakeMaterials failed ({rc})")
    return size, start, tex


def scene_from_meshes(meshes: Sequence[Mesh], materials: Sequence[dict], lights: Sequence[dict], position, look_at, up, fov: float,
                      width: int, height: int, samples: int = 1, name: str = "mesh scene") -> Scene:
    """The tail of parseAndRender up to the builders (render.cpp:1051-1309): camera, geometry, materials, lights."""
    tl, lr, tb, inv = set_camera(position, look_at, up, fov, width, height)
    eye = np.zeros(4, np.float32)
    eye[:3] = np.asarray(position, np.float32)
    vertex, tri_index, tri_material, tri_uv, tri_normal = mesh_arrays(meshes, eye)
    mat_size, mat_start, textures = bake_materials(materials)
    ltype, lpos, ldir, lcol, lrad, lhalf = light_arrays(lights)
    return Scene(width=width, height=height, eye=eye, eye_to_top_left=tl, left_to_right=lr, top_to_bottom=tb, pixel_size_inv=inv,
                 sample_count=samples, vertex=vertex, tri_index=tri_index, tri_material=tri_material, tri_uv=tri_uv, tri_normal=tri_normal,
                 mat_size=mat_size, mat_start=mat_start, textures=textures, light_type=ltype, light_pos=lpos, light_dir=ldir,
                 light_col=lcol, light_radius=lrad, light_half_att=lhalf, name=name)


PROJ_SPHERICAL, PROJ_CYLINDRICAL, PROJ_FLAT, PROJ_CUBIC, PROJ_FRONTAL, PROJ_SPATIAL, PROJ_UVW, PROJ_SHRINKWRAP, PROJ_VOLUMESHADER = 0, 1, 2, 3, 4, 5, 6, 7, 10


def project_uv(projection: int, point, normal, offset=(0.0, 0.0), length=(1.0, 1.0), tile: bool = True, start=(0.0, 0.0)):
    """ShdProjectPoint (render.cpp:495-673).  Returns ((u, v) as float32, inside)."""
    uv = np.asarray(start, np.float32).copy()
    inside = _lib().rtHipProjectUv(int(projection), _f3(point), _f3(normal), float(offset[0]), float(offset[1]), float(length[0]), float(length[1]),
                                   1 if tile else 0, uv.ctypes.data_as(C.POINTER(C.c_float)))
    return uv, bool(inside)


def read_image(path: str) -> np.ndarray:
    """PPM (P6 / P3) or BMP (24 / 32 bit) -> [h,w,3] uint8, top row first."""
    w, h, px = C.c_uint32(), C.c_uint32(), C.c_void_p()
    rc = _lib().rtHipImageRead(path.encode(), C.byref(w), C.byref(h), C.byref(px))
    if rc != 0:
        raise OSError(f"rtHipImageRead({path}) failed ({rc})")
    out = np.empty((h.value, w.value, 4), np.uint8)
    C.memmove(out.ctypes.data, px.value, out.nbytes)
    R.lib().rtHipFree(px)
    return np.ascontiguousarray(out[:, :, :3])


def read_obj(path: str):
    """A Wavefront OBJ (+ MTL) as (Mesh, materials): the mesh in rtHipMesh's shape, the materials as the dicts bake_materials takes
    (render.cpp:1136-1309's channel rules then apply): colour = map_Kd image or Kd; transparency on when d < 1 or map_d is given
    (1x1 of 1 - d without an image); reflection on when `refl` or map_refl is given; bump from map_bump; luminance from map_Ke or Ke."""
    d = _ObjData()
    rc = _lib().rtHipObjRead(path.encode(), C.byref(d))
    if rc != 0:
        raise OSError(f"rtHipObjRead({path}) failed ({rc})")
    try:
        def arr(ptr, shape, dtype):
            if not ptr:
                return None
            out = np.empty(shape, dtype)
            C.memmove(out.ctypes.data, ptr, out.nbytes)
            return out
        n = d.polygonCount
        mesh = Mesh(points=arr(d.points, (d.pointCount, 4), np.float32)[:, :3], polygons=arr(d.polygons, (n, 4), np.int32),
                    corner_normals=None if not d.cornerNormals else arr(d.cornerNormals, (n, 4, 4), np.float32)[:, :, :3],
                    corner_uv=arr(d.cornerUv, (n, 4, 2), np.float32), polygon_material=arr(d.polygonMaterial, (n,), np.int32))
        materials = []
        for i in range(d.materialCount):
            m = d.materials[i]
            maps = [bytes(m.map[c]).split(b"\0", 1)[0].decode() for c in range(5)]
            byte3 = lambda v: np.clip(np.round(np.asarray(v, np.float64) * 255.0), 0, 255).astype(np.uint8).reshape(1, 1, 3)
            mat = dict(name=m.name.decode(), rgb=tuple(m.kd), brightness=1.0)
            mat["color"] = read_image(maps[0]) if maps[0] else True
            if maps[1]:
                mat["reflection"] = read_image(maps[1])
            elif m.hasReflect:
                mat["reflection"] = byte3([m.reflect] * 3)
            if maps[2]:
                mat["transparency"] = read_image(maps[2])
            elif m.dissolve < 1.0:
                mat["transparency"] = byte3([1.0 - m.dissolve] * 3)
            if maps[3]:
                mat["bump"] = read_image(maps[3])
            if maps[4]:
                mat["luminance"] = read_image(maps[4])
            elif m.hasKe:
                mat["luminance"] = byte3(m.ke)
            materials.append(mat)
        return mesh, materials
    finally:
        _lib().rtHipObjFree(C.byref(d))


def planes_to_rgb8(r, g, b, low_byte_compat: bool = False) -> np.ndarray:
    """[H,W] u16 planes -> [H,W,3] uint8 (value / 256, render.cpp:1379-1382)."""
    h, w = r.shape
    planes = [np.ascontiguousarray(p, np.uint16) for p in (r, g, b)]
    out = np.zeros((h, w, 3), np.uint8)
    _lib().rtHipPlanesToRgb8(w, h, _p(planes[0]), _p(planes[1]), _p(planes[2]), _p(out), 1 if low_byte_compat else 0)
    return out


def write_bmp(path: str, r, g, b, low_byte_compat: bool = False) -> None:
    h, w = r.shape
    planes = [np.ascontiguousarray(p, np.uint16) for p in (r, g, b)]
    rc = _lib().rtHipWriteBmp(path.encode(), w, h, _p(planes[0]), _p(planes[1]), _p(planes[2]), 1 if low_byte_compat else 0)
    if rc != 0:
        raise OSError(f"rtHipWriteBmp({path}) failed ({rc})")


def write_ppm(path: str, r, g, b) -> None:
    h, w = r.shape
    planes = [np.ascontiguousarray(p, np.uint16) for p in (r, g, b)]
    rc = _lib().rtHipWritePpm(path.encode(), w, h, _p(planes[0]), _p(planes[1]), _p(planes[2]))
    if rc != 0:
        raise OSError(f"rtHipWritePpm({path}) failed ({rc})")
