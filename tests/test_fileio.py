"""File input of the headless front-end (rt_fileio.cpp): OBJ / MTL, PPM / BMP, and the projected UVs of ShdProjectPoint
(reference source/render.cpp:495-673) against an independent numpy restatement.  Nothing of render.cpp compiles without the Maxon
SDK and the reference holds no fixtures: parity unpinned (libm acos / atan / sin / cos are this machine's)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from opencl_render_amd import frontend as F

DATA = os.path.join(ROOT, "tests", "data")


def test_obj_reader_gives_the_polygon_objects_the_front_end_takes():
    mesh, mats = F.read_obj(os.path.join(DATA, "scene.obj"))
    assert mesh.points.shape == (22, 3) and mesh.polygons.shape == (13, 4)     # 1 + 5 quads, 4 triangles, a pentagon fanned into 3
    assert [m["name"] for m in mats] == ["floor", "glass", "rough", "lamp"]     # in order of first use / definition
    pol = mesh.polygons
    assert pol[0].tolist() == [0, 1, 2, 3] and pol[1].tolist() == [4, 5, 6, 7]                   # quads keep a,b,c,d
    assert pol[6].tolist() == [12, 13, 16, 16] and pol[9].tolist() == [15, 12, 16, 16]           # triangles: c == d; negative indices resolved
    assert pol[10].tolist() == [17, 18, 19, 19] and pol[12].tolist() == [17, 20, 21, 21]         # the pentagon's fan
    assert mesh.polygon_material.tolist() == [0] + [1] * 5 + [2] * 4 + [3] * 3
    assert np.array_equal(mesh.corner_uv[0], np.float32([[0, 0], [3, 0], [3, 3], [0, 3]]))
    assert np.array_equal(mesh.corner_uv[6], np.float32([[0, 0], [3, 0], [0.5, 1], [0.5, 1]]))   # (vt 1, 2, 5; the triangle's 4th corner repeats the 3rd)
    assert np.array_equal(mesh.corner_uv[1], np.zeros((4, 2), np.float32))                       # a face without vt
    assert np.array_equal(mesh.corner_normals[0], np.float32([[0, 1, 0]] * 4)) and not mesh.corner_normals[1].any()
    floor, glass, rough, lamp = mats
    assert floor["color"].shape == (8, 8, 3) and tuple(np.float32(floor["rgb"])) == tuple(np.float32([0.8, 0.8, 0.7]))
    assert glass["transparency"].reshape(-1).tolist() == [153] * 3 and glass["reflection"].reshape(-1).tolist() == [77] * 3  # 1 - d, refl (0.3f * 255 = 76.500003)
    assert rough["bump"].shape == (4, 6, 3) and lamp["luminance"].reshape(-1).tolist() == [255, 229, 153]  # (0.9f * 255 = 229.49998)


def test_image_readers_ppm_and_bmp(tmp_path):
    tex = F.read_image(os.path.join(DATA, "checker.ppm"))
    raw = open(os.path.join(DATA, "checker.ppm"), "rb").read()
    assert tex.shape == (8, 8, 3) and tex.tobytes() == raw[-192:]
    assert tex[0, 0].tolist() == [255, 40, 40]
    bump = F.read_image(os.path.join(DATA, "bump.bmp"))
    assert bump.shape == (4, 6, 3) and (bump[:, :, 0] == bump[:, :, 1]).all()
    # a round trip through the library's own BMP writer (bottom-up rows, BGR, 4-byte row padding) and a plain PPM
    img = np.random.default_rng(1).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    planes = [img[:, :, c].astype(np.uint16) * 256 for c in range(3)]
    F.write_bmp(str(tmp_path / "a.bmp"), *planes)
    assert np.array_equal(F.read_image(str(tmp_path / "a.bmp")), img)
    (tmp_path / "b.ppm").write_text("P3\n# plain\n2 1\n15\n15 0 3  1 2 15\n")
    assert F.read_image(str(tmp_path / "b.ppm")).tolist() == [[[255, 0, 51], [17, 34, 255]]]
    (tmp_path / "c.ppm").write_bytes(b"P6\n2 2\n255\n\x00\x01")  # truncated
    with pytest.raises(OSError):
        F.read_image(str(tmp_path / "c.ppm"))
    with pytest.raises(OSError):
        F.read_image(str(tmp_path / "missing.bmp"))


def _project_numpy(proj, p, n, ox, oy, lenx, leny, tile, start):
    """Independent restatement of ShdProjectPoint (render.cpp:495-673) in float64."""
    px, py, pz = (np.float64(v) for v in p)
    lenxinv = 1.0 / lenx if lenx != 0 else 0.0
    lenyinv = 1.0 / leny if leny != 0 else 0.0
    u, v = np.float64(start[0]), np.float64(start[1])
    sq = np.sqrt(px * px + pz * pz)
    wrap_u = None
    if proj == F.PROJ_VOLUMESHADER:
        return np.float32([px, py]), True
    if proj in (F.PROJ_FRONTAL, F.PROJ_UVW):
        pass
    elif proj == F.PROJ_SHRINKWRAP:
        if sq == 0:
            u, v = 0.0, (0.0 if py > 0 else 1.0)
        else:
            u = np.arccos(px / sq) / (2 * np.pi)
            if pz < 0: u = 1.0 - u
            v = 0.5 - np.arctan(py / sq) / np.pi
        sn, cs = np.sin(u * 2 * np.pi), np.cos(u * 2 * np.pi)
        u, v = (0.5 + 0.5 * cs * v - ox) * lenxinv, (0.5 + 0.5 * sn * v - oy) * lenyinv
    elif proj == F.PROJ_CYLINDRICAL:
        if sq == 0:
            u = 0.0
        else:
            u = np.arccos(px / sq) / (2 * np.pi)
            if pz < 0: u = 1.0 - u
            u -= ox
            if lenx > 0 and u < 0: u += 1.0
            elif lenx < 0 and u > 0: u -= 1.0
            u *= lenxinv
        v = -(py * 0.5 + oy) * lenyinv
    elif proj in (F.PROJ_FLAT, F.PROJ_SPATIAL):
        u, v = (px * 0.5 - ox) * lenxinv, -(py * 0.5 + oy) * lenyinv
    elif proj == F.PROJ_CUBIC:
        ax, ay, az = (abs(np.float64(c)) for c in n)
        axis = (0 if ax > az else 2) if ax > ay else (1 if ay > az else 2)
        if axis == 0:
            u = ((-pz if n[0] < 0 else pz) * 0.5 - ox) * lenxinv
            v = -(py * 0.5 + oy) * lenyinv
        elif axis == 1:
            v = ((pz if n[1] < 0 else -pz) * 0.5 - oy) * lenyinv
            u = (px * 0.5 - ox) * lenxinv
        else:
            u = ((px if n[2] < 0 else -px) * 0.5 - ox) * lenxinv
            v = -(py * 0.5 + oy) * lenyinv
    else:  # spherical and every unknown number
        if sq == 0:
            u, v = 0.0, (0.5 if py > 0 else -0.5)
        else:
            u = np.arccos(px / sq) / (2 * np.pi)
            if pz < 0: u = 1.0 - u
            u -= ox
            if lenx > 0 and u < 0: u += 1.0
            elif lenx < 0 and u > 0: u -= 1.0
            u *= lenxinv
            v = 0.5 + np.arctan(py / sq) / np.pi
        v = -(v - oy) * lenyinv
    return np.float32([u, v]), bool(tile or (0 <= u <= 1 and 0 <= v <= 1))


def test_projected_uvs_match_a_numpy_restatement():
    rng = np.random.default_rng(11)
    projs = [F.PROJ_SPHERICAL, F.PROJ_CYLINDRICAL, F.PROJ_FLAT, F.PROJ_CUBIC, F.PROJ_FRONTAL, F.PROJ_SPATIAL, F.PROJ_UVW, F.PROJ_SHRINKWRAP,
             F.PROJ_VOLUMESHADER, 99]
    points = [rng.uniform(-2, 2, 3) for _ in range(40)] + [np.array([0.0, 1.0, 0.0]), np.array([0.0, -2.0, 0.0]), np.array([1.0, 0.0, 0.0]),
                                                           np.array([-1.0, 0.3, -0.0]), np.array([0.5, 0.5, -1e-30])]
    worst = 0.0
    for proj in projs:
        for p in points:
            p = np.float32(p)
            n = np.float32(rng.uniform(-1, 1, 3))
            for ox, oy, lenx, leny, tile in [(0, 0, 1, 1, True), (0.25, -0.1, 0.5, 2.0, False), (0.9, 0.2, -1.0, 0.0, False), (0, 0, 0, 1, True)]:
                got, inside = F.project_uv(proj, p, n, (ox, oy), (lenx, leny), tile, start=(0.125, -7.0))
                want, want_inside = _project_numpy(proj, p, n, np.float32(ox), np.float32(oy), np.float32(lenx), np.float32(leny), tile, (0.125, -7.0))
                assert inside == want_inside, (proj, p, ox, oy, lenx, leny)
                # same formulas in double on both sides; only libm's last bit may differ before the cast to float
                assert np.allclose(got, want, rtol=0, atol=2e-7 * max(1.0, float(np.abs(want).max()))), (proj, p, got, want)
                worst = max(worst, float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()))
    # the cubic projection follows the dominant axis of the normal, ties to the later axis (render.cpp:600-611)
    assert F.project_uv(F.PROJ_CUBIC, (1, 2, 3), (1, 1, 0))[0].tolist() == [0.5, -1.5]          # |x| > |y| fails -> y vs z -> y axis: v = -z/2
    assert F.project_uv(F.PROJ_CUBIC, (1, 2, 3), (0, 1, 1))[0].tolist() == [-0.5, -1.0]         # |y| > |z| fails -> z axis, v.z > 0
    assert F.project_uv(F.PROJ_FRONTAL, (1, 2, 3), (0, 0, 1), start=(0.25, 0.5))[0].tolist() == [0.25, 0.5]  # untouched, as in the reference
