"""Scene container in the reference ABI's memory layouts, plus seeded synthetic scenes.

The arrays are exactly what ``RaytraceAll`` receives (reference ``source/opencl/raytrace.h:58-106``):
``float3``/``int3`` are 16-byte rows (lane 3 is padding), ``float2``/``uint2`` 8 bytes, ``uchar3`` 4 bytes
(``source/3rdparty/opencl-1.2/include/CL/cl_platform.h:501,725,1025``).  The synthetic generator follows the
recipe of SURVEY.md section 8(d): a non-indexed triangle soup inside the view frustum of a camera at the origin.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

GRID_DIV = 256  # reference trianglelist.h:110
CH_COLOR, CH_REFLECTION, CH_TRANSPARENCY, CH_BUMP, CH_LUMINANCE, CH_COUNT = 0, 1, 2, 3, 4, 5  # raytrace_opencl.h:14-22
# raytrace_opencl.h:1-12
LIGHT_OMNI, LIGHT_SPOT, LIGHT_SPOTRECT, LIGHT_DISTANT, LIGHT_PARALLEL = 0, 1, 2, 3, 4
LIGHT_PARSPOT, LIGHT_PARSPOTRECT, LIGHT_TUBE, LIGHT_AREA, LIGHT_PHOTOMETRIC = 5, 6, 7, 8, 9


def _f4(v) -> np.ndarray:
    out = np.zeros(4, np.float32)
    out[:3] = np.asarray(v, np.float32)
    return out


@dataclass
class Scene:
    width: int
    height: int
    eye: np.ndarray
    eye_to_top_left: np.ndarray
    left_to_right: np.ndarray
    top_to_bottom: np.ndarray
    pixel_size_inv: float
    sample_count: int
    vertex: np.ndarray          # [V,4] f32
    tri_index: np.ndarray       # [T,4] i32
    tri_material: np.ndarray    # [T] i32 (-1 = none)
    tri_uv: np.ndarray          # [3T,2] f32
    tri_normal: np.ndarray      # [3T,4] f32
    mat_size: np.ndarray        # [5M,2] u32
    mat_start: np.ndarray       # [5M+1] i32
    textures: np.ndarray        # [texels,4] u8
    light_type: np.ndarray      # [L] i32
    light_pos: np.ndarray       # [L,4] f32
    light_dir: np.ndarray       # [L,4] f32
    light_col: np.ndarray       # [L,4] f32
    light_radius: np.ndarray    # [L] f32
    light_half_att: np.ndarray  # [L] f32
    # acceleration structures (filled by the builders)
    cam_start: Optional[np.ndarray] = None   # [P] u32
    cam_end: Optional[np.ndarray] = None     # [P] u32
    cam_list: Optional[np.ndarray] = None    # [n] u32
    box_min: Optional[np.ndarray] = None     # [257,4] f32
    grid_start: Optional[np.ndarray] = None  # [256^3+1] u32
    grid_list: Optional[np.ndarray] = None   # [n] u32
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def pixels(self) -> int:
        return self.width * self.height

    @property
    def triangle_count(self) -> int:
        return int(self.tri_index.shape[0])

    @property
    def vertex_count(self) -> int:
        return int(self.vertex.shape[0])

    @property
    def material_count(self) -> int:
        return int(self.mat_size.shape[0] // CH_COUNT)

    @property
    def light_count(self) -> int:
        return int(self.light_type.shape[0])

    def sum_candidates(self) -> int:
        """Sum over pixels of K_p = End[p]-Start[p]."""
        return int((self.cam_end.astype(np.int64) - self.cam_start.astype(np.int64)).clip(min=0).sum())


# ---------------------------------------------------------------------------------------------------------------
# materials
# ---------------------------------------------------------------------------------------------------------------

def _channel_image(spec) -> Optional[np.ndarray]:
    """None -> absent (0x0); (r,g,b) bytes -> 1x1; [H,W,3] uint8 -> image."""
    if spec is None:
        return None
    arr = np.asarray(spec)
    if arr.ndim == 1:
        return arr.astype(np.uint8).reshape(1, 1, 3)
    assert arr.ndim == 3 and arr.shape[2] == 3
    return arr.astype(np.uint8)


def pack_materials(materials: Sequence[dict]):
    """materials: dicts with optional keys color/reflection/transparency/bump/luminance.
    Returns (mat_size[5M,2] u32, mat_start[5M+1] i32, textures[texels,4] u8)."""
    keys = ["color", "reflection", "transparency", "bump", "luminance"]
    sizes = np.zeros((CH_COUNT * len(materials), 2), np.uint32)
    starts = np.zeros(CH_COUNT * len(materials) + 1, np.int32)
    texels = []
    cursor = 0
    for m, mat in enumerate(materials):
        for c, key in enumerate(keys):
            img = _channel_image(mat.get(key))
            starts[CH_COUNT * m + c] = cursor
            if img is not None:
                h, w = img.shape[:2]
                sizes[CH_COUNT * m + c] = (w, h)
                px = np.zeros((h * w, 4), np.uint8)
                px[:, :3] = img.reshape(-1, 3)
                texels.append(px)
                cursor += h * w
    starts[-1] = cursor  # last entry = total texels (reference render.cpp:1306, relied on at raytrace.c:441)
    tex = np.concatenate(texels) if texels else np.zeros((0, 4), np.uint8)
    return sizes, starts, np.ascontiguousarray(tex)


def pack_lights(lights: Sequence[dict]):
    n = len(lights)
    ltype = np.zeros(n, np.int32)
    pos = np.zeros((n, 4), np.float32)
    direction = np.zeros((n, 4), np.float32)
    col = np.zeros((n, 4), np.float32)
    radius = np.zeros(n, np.float32)
    half = np.zeros(n, np.float32)
    for i, l in enumerate(lights):
        ltype[i] = l["type"]
        pos[i, :3] = l.get("pos", (0, 0, 0))
        direction[i, :3] = l.get("dir", (0, 0, 1))
        col[i, :3] = l.get("col", (1, 1, 1))
        radius[i] = l.get("radius", 0.52)          # sun angle used for every light (reference render.cpp:961)
        half[i] = l.get("half_att", np.inf)        # half-attenuation distance is infinite in reference scenes (render.cpp:976)
    return ltype, pos, direction, col, radius, half


# ---------------------------------------------------------------------------------------------------------------
# synthetic soups
# ---------------------------------------------------------------------------------------------------------------

def make_soup(width: int, height: int, triangles: int, edge: float, seed: int = 12345, samples: int = 1,
              materials: Optional[Sequence[dict]] = None, lights: Optional[Sequence[dict]] = None,
              random_uv: bool = False, material_ids: Optional[np.ndarray] = None, smooth_normals: bool = False,
              depth=(2.0, 4.0), name: str = "") -> Scene:
    """Seeded triangle soup: non-indexed (V = 3T), centres uniform in the view frustum at depth z in `depth`,
    vertex offsets uniform in +-edge/2, flat normals = normalize(cross(ab, ac))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    aspect = np.float32(height) / np.float32(width)
    z = rng.uniform(depth[0], depth[1], triangles).astype(np.float32)
    cx = (rng.uniform(-0.5, 0.5, triangles).astype(np.float32) * z)
    cy = (rng.uniform(-0.5, 0.5, triangles).astype(np.float32) * aspect * z)
    centre = np.stack([cx, cy, z], 1)
    off = rng.uniform(-0.5, 0.5, (triangles, 3, 3)).astype(np.float32) * np.float32(edge)
    pts = (centre[:, None, :] + off).astype(np.float32)  # [T,3,3]

    vertex = np.zeros((3 * triangles, 4), np.float32)
    vertex[:, :3] = pts.reshape(-1, 3)
    tri_index = np.zeros((triangles, 4), np.int32)
    tri_index[:, 0] = 3 * np.arange(triangles)
    tri_index[:, 1] = tri_index[:, 0] + 1
    tri_index[:, 2] = tri_index[:, 0] + 2

    ab = pts[:, 1] - pts[:, 0]
    ac = pts[:, 2] - pts[:, 0]
    n = np.cross(ab, ac).astype(np.float32)
    ln = np.sqrt((n * n).sum(1, dtype=np.float32)).astype(np.float32)
    ln[ln == 0] = 1
    n = (n / ln[:, None]).astype(np.float32)
    tri_normal = np.zeros((3 * triangles, 4), np.float32)
    if smooth_normals:
        jitter = rng.uniform(-0.3, 0.3, (triangles, 3, 3)).astype(np.float32)
        nn = n[:, None, :] + jitter
        nn = nn / np.sqrt((nn * nn).sum(2, keepdims=True)).astype(np.float32)
        tri_normal[:, :3] = nn.reshape(-1, 3).astype(np.float32)
    else:
        tri_normal[:, :3] = np.repeat(n, 3, axis=0)

    tri_uv = np.zeros((3 * triangles, 2), np.float32)
    if random_uv:
        tri_uv[:] = rng.uniform(-1.5, 2.5, (3 * triangles, 2)).astype(np.float32)

    if materials is None:
        # colour 1x1 white, refl/transp/bump/lum 1x1 black
        materials = [dict(color=(255, 255, 255), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(0, 0, 0))]
    if lights is None:
        lights = [dict(type=LIGHT_DISTANT, dir=(0.3, -0.8, 0.5), col=(1, 1, 1), radius=0.52)]
    mat_size, mat_start, textures = pack_materials(materials)
    if material_ids is None:
        if len(materials) == 0:
            material_ids = np.full(triangles, -1, np.int32)
        else:
            material_ids = rng.integers(0, len(materials), triangles).astype(np.int32)
    ltype, lpos, ldir, lcol, lrad, lhalf = pack_lights(lights)

    return Scene(
        width=width, height=height,
        eye=_f4((0, 0, 0)), eye_to_top_left=_f4((-0.5, 0.5 * float(aspect), 1.0)),
        left_to_right=_f4((1.0 / width, 0, 0)), top_to_bottom=_f4((0, -1.0 / width, 0)),
        pixel_size_inv=float(width), sample_count=samples,
        vertex=vertex, tri_index=tri_index, tri_material=np.ascontiguousarray(material_ids, np.int32),
        tri_uv=tri_uv, tri_normal=tri_normal,
        mat_size=mat_size, mat_start=mat_start, textures=textures,
        light_type=ltype, light_pos=lpos, light_dir=ldir, light_col=lcol, light_radius=lrad, light_half_att=lhalf,
        name=name or f"soup_{width}x{height}_{triangles}", meta=dict(seed=seed, edge=edge),
    )


def primary_only_material(texture_side: int = 0, seed: int = 7) -> dict:
    """colour black, luminance white (or a seeded texture) => out = luminance, weight 0 =>
    no secondary rays are spawned (raytrace_opencl.c:639-667)."""
    if texture_side:
        rng = np.random.Generator(np.random.PCG64(seed))
        lum = rng.integers(0, 256, (texture_side, texture_side, 3)).astype(np.uint8)
    else:
        lum = (255, 255, 255)
    return dict(color=(0, 0, 0), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=lum)
