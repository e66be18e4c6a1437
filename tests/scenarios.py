"""Seeded parity scenarios shared by the golden-fixture generator, the oracle tests and the GPU parity tests.

Each scenario is small enough for the single-thread oracle to finish in well under a second and is aimed at one
part of the reference kernel (raytrace_opencl.c line ranges in the comments).  Inputs are deterministic functions
of the seed (numpy PCG64); the golden fixtures nevertheless store the inputs themselves (tests/golden/*.npz).
"""
from __future__ import annotations

import numpy as np

from opencl_render_amd import scene as S


def _tex(seed, side, lo=0, hi=256):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(lo, hi, (side, side, 3)).astype(np.uint8)


def _lambert(**kw):
    m = dict(color=(255, 255, 255), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(0, 0, 0))
    m.update(kw)
    return m


def lambert_distant():
    """SURVEY 8d config 3 in miniature: white Lambert, one distant light, shadow ray + diffuse bounce (:589-606,:664-683)."""
    return S.make_soup(64, 48, 1500, 0.06, seed=11, samples=2, name="lambert_distant")


def primary_only():
    """luminance-only material with a texture, no lights => no secondary rays (:514-528,:639-641)."""
    return S.make_soup(96, 64, 3000, 0.05, seed=12, samples=1, materials=[S.primary_only_material(16)], lights=[],
                       random_uv=True, name="primary_only")


def all_light_types():
    """One light of every type 0..9 plus an unknown type (:566-607): sphere sampling, omni's zero contribution."""
    lights = []
    rng = np.random.Generator(np.random.PCG64(5))
    for t in list(range(10)) + [42]:
        lights.append(dict(type=t, pos=tuple(rng.uniform(-1, 1, 3) + np.array([0, 0, 1.0])), dir=tuple(rng.uniform(-1, 1, 3)),
                           col=tuple(rng.uniform(0.1, 0.6, 3)), radius=float(rng.uniform(0.05, 0.6))))
    return S.make_soup(48, 48, 1200, 0.08, seed=13, samples=1, lights=lights, name="all_light_types")


def spot_finite_range():
    """Positional lights: finite shadow-ray range => end-cell logic of the DDA (:356-362,:380-381), finite half
    attenuation => pow(0.5, d/h) (:631)."""
    lights = [dict(type=S.LIGHT_SPOT, pos=(0.4, 0.6, 1.5), col=(0.9, 0.8, 0.7), radius=0.2, half_att=2.0),
              dict(type=S.LIGHT_AREA, pos=(-0.8, -0.3, 2.5), col=(0.3, 0.5, 0.9), radius=0.05, half_att=0.75)]
    return S.make_soup(64, 40, 2000, 0.07, seed=14, samples=2, lights=lights, name="spot_finite_range")


def no_material():
    """materialId -1 (:231,:550: zero texture => black, no bounces, no bump) on half of the triangles, mixed with a lit
    textured material on the others: black pixels where a -1 triangle is nearest, and -1 triangles still occlude
    (shadow rays :612-626 treat them as opaque, bounce rays stop on them)."""
    mats = [_lambert(color=_tex(21, 6, 60, 256), bump=_tex(22, 5))]
    sc = S.make_soup(40, 40, 500, 0.3, seed=15, samples=2, materials=mats, random_uv=True, name="no_material")
    rng = np.random.Generator(np.random.PCG64(15))
    sc.tri_material[rng.random(sc.triangle_count) < 0.5] = -1
    return sc


def mixed_materials_textured():
    """Five materials with images on every channel, wrapping UVs (positive_modf :25-28), material -1 mixed in,
    smooth (un-normalised Phong) normals (:221-229), bump maps (:231-261)."""
    mats = [
        _lambert(color=_tex(1, 8)),
        _lambert(color=_tex(2, 5), bump=_tex(3, 7)),
        _lambert(color=(200, 180, 160), luminance=_tex(4, 4, 0, 90)),
        _lambert(color=_tex(5, 6), reflection=_tex(6, 3, 0, 128)),
        _lambert(color=(255, 255, 255), transparency=_tex(7, 4, 100, 256), bump=_tex(8, 9)),
    ]
    sc = S.make_soup(64, 64, 1800, 0.09, seed=16, samples=2, materials=mats, random_uv=True, smooth_normals=True,
                     lights=[dict(type=S.LIGHT_DISTANT, dir=(0.2, -0.7, 0.6), col=(0.9, 0.9, 0.8)),
                             dict(type=S.LIGHT_TUBE, pos=(0.5, 0.5, 0.5), col=(0.4, 0.3, 0.2), radius=0.1)],
                     name="mixed_materials_textured")
    rng = np.random.Generator(np.random.PCG64(99))
    sc.tri_material[rng.random(sc.triangle_count) < 0.1] = -1
    return sc


def mirror_hall():
    """Highly reflective + transparent big triangles: mirror recursion to depth 12, see-through continuation rays
    that stay 'fromCamera' (:707-722), ring-full checks (:682,:704,:721), transparent shadow chains (:609-626)."""
    mats = [
        _lambert(color=(255, 255, 255), reflection=(230, 230, 230)),
        _lambert(color=(255, 250, 245), transparency=(200, 210, 220)),
        _lambert(color=(255, 255, 255), reflection=(120, 120, 120), transparency=(120, 120, 120)),
        _lambert(color=(240, 240, 240)),
    ]
    return S.make_soup(48, 36, 300, 0.9, seed=17, samples=2, materials=mats,
                       lights=[dict(type=S.LIGHT_DISTANT, dir=(0.1, -0.9, 0.3), col=(1, 1, 1)),
                               dict(type=S.LIGHT_PHOTOMETRIC, pos=(0, 0.5, 3.0), col=(0.5, 0.5, 0.6), radius=0.3)],
                       name="mirror_hall")


def degenerate_and_outside():
    """Zero-area triangles (NaN barycentrics), a camera placed outside the scene's bounding box so secondary rays
    start outside / leave the grid (BindInCube early returns :265-322), a light with negative colour and one bright
    enough to saturate (:726-741)."""
    sc = S.make_soup(56, 40, 900, 0.12, seed=18, samples=3,
                     lights=[dict(type=S.LIGHT_DISTANT, dir=(-0.5, -0.5, 0.2), col=(3.0, 2.0, 40.0)),
                             dict(type=S.LIGHT_PARALLEL, dir=(0.5, 0.1, -0.4), col=(-0.5, 0.2, 0.1))],
                     name="degenerate_and_outside")
    # collapse every 7th triangle to a point and every 11th to a segment
    v = sc.vertex.reshape(-1, 3, 4)
    v[::7, 1] = v[::7, 0]
    v[::7, 2] = v[::7, 0]
    v[::11, 2] = v[::11, 1]
    return sc


def sparse_many_samples():
    """Few triangles, many empty pixels and long empty DDA walks; S=5 exercises per-sample truncation (:728-740)."""
    return S.make_soup(80, 60, 150, 0.5, seed=19, samples=5, name="sparse_many_samples")


def odd_size_multi_tile():
    """Image larger than one 128x128 tile with ragged edges (tile partition)."""
    return S.make_soup(200, 150, 2500, 0.08, seed=20, samples=1, name="odd_size_multi_tile")


ALL = [lambert_distant, primary_only, all_light_types, spot_finite_range, no_material, mixed_materials_textured,
       mirror_hall, degenerate_and_outside, sparse_many_samples, odd_size_multi_tile]


def by_name(name):
    for f in ALL:
        if f.__name__ == name:
            return f
    raise KeyError(name)
