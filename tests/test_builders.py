"""CPU tests of the host builders (counterparts of trianglelist.cpp:520-626,655-737): structural invariants and the
property the kernel relies on -- the per-pixel candidate list never loses the triangle a brute-force search finds."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from opencl_render_amd import raytrace as R, scene as S


@pytest.fixture(scope="module")
def built():
    sc = S.make_soup(72, 56, 1200, 0.12, seed=77)
    R.build_lists(sc, threads=3)
    return sc


def test_camera_list_structure(built):
    sc = built
    assert sc.cam_start.shape == (sc.pixels,) and sc.cam_end.shape == (sc.pixels,)
    assert (sc.cam_end >= sc.cam_start).all() and sc.cam_end.max() <= len(sc.cam_list)
    assert sc.cam_list.max() < sc.triangle_count
    for p in np.random.default_rng(0).integers(0, sc.pixels, 300):
        seg = sc.cam_list[sc.cam_start[p]:sc.cam_end[p]]
        assert (np.diff(seg.astype(np.int64)) > 0).all()  # ascending triangle ids, no duplicates (trianglelist.cpp:161,565)
    # neighbour de-duplication (trianglelist.cpp:580-613): equal neighbours alias the same range
    w = sc.width
    for p in range(1, sc.pixels):
        if p % w and sc.cam_end[p] - sc.cam_start[p] == sc.cam_end[p - 1] - sc.cam_start[p - 1]:
            a = sc.cam_list[sc.cam_start[p]:sc.cam_end[p]]
            b = sc.cam_list[sc.cam_start[p - 1]:sc.cam_end[p - 1]]
            if np.array_equal(a, b):
                assert sc.cam_start[p] == sc.cam_start[p - 1]


def test_thread_count_does_not_change_the_lists(built):
    sc2 = S.make_soup(72, 56, 1200, 0.12, seed=77)
    R.build_lists(sc2, threads=1)
    for k in ("cam_start", "cam_end", "cam_list", "box_min", "grid_start", "grid_list"):
        assert np.array_equal(getattr(built, k), getattr(sc2, k)), k


def test_grid_structure(built):
    sc = built
    assert sc.grid_start.shape == (256 ** 3 + 1,) and sc.grid_start[0] == 0
    assert (np.diff(sc.grid_start.astype(np.int64)) >= 0).all() and sc.grid_start[-1] == len(sc.grid_list)
    for w in range(3):
        assert (np.diff(sc.box_min[:, w]) >= 0).all()
        srt = np.sort(sc.vertex[:, w])
        # plane 0 is the minimum; plane 256 is the midpoint of the two largest values (trianglelist.cpp:669-673)
        assert sc.box_min[0, w] == srt[0] and sc.box_min[256, w] == (srt[-1] + srt[-2]) / np.float32(2)
    # every triangle is listed in the cell of its first vertex (FillCube's seed cell, trianglelist.cpp:455-459)
    L = O.oracle()
    for t in np.random.default_rng(1).integers(0, sc.triangle_count, 200):
        a = np.ascontiguousarray(sc.vertex[sc.tri_index[t, 0], :3])
        cell = (C.c_int * 3)()
        L.rt_oracle_box_address(256, sc.box_min.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.POINTER(C.c_float)), cell)
        cid = cell[0] + 256 * cell[1] + 65536 * cell[2]
        seg = sc.grid_list[sc.grid_start[cid]:sc.grid_start[cid + 1]]
        assert t in seg
        assert (np.diff(seg.astype(np.int64)) > 0).all()


def test_candidate_lists_are_conservative_for_pixel_centres(built):
    """Brute force over ALL triangles vs the pixel's candidate list, for rays through pixel centres."""
    sc = built
    L = O.oracle()
    fp = C.POINTER(C.c_float)
    rng = np.random.default_rng(2)
    eye = np.ascontiguousarray(sc.eye[:3])
    tri_pts = sc.vertex[sc.tri_index[:, :3], :3]  # [T,3,3]
    checked = 0
    for p in rng.integers(0, sc.pixels, 400):
        x, y = p % sc.width, p // sc.width
        d = (sc.eye_to_top_left[:3] + sc.left_to_right[:3] * np.float32(x + 0.5) + sc.top_to_bottom[:3] * np.float32(y + 0.5)).astype(np.float32)
        d = np.ascontiguousarray(d)
        hits = []
        for t in range(sc.triangle_count):
            tt, l1, l2 = C.c_float(), C.c_float(), C.c_float()
            a, b, c = (np.ascontiguousarray(tri_pts[t, k]) for k in range(3))
            if L.rt_oracle_ray_triangle(eye.ctypes.data_as(fp), d.ctypes.data_as(fp), 0.0, float("inf"), a.ctypes.data_as(fp),
                                        b.ctypes.data_as(fp), c.ctypes.data_as(fp), C.byref(tt), C.byref(l1), C.byref(l2)):
                hits.append(t)
        cand = set(sc.cam_list[sc.cam_start[p]:sc.cam_end[p]].tolist())
        assert set(hits) <= cand, (p, hits, cand)
        checked += len(hits)
    assert checked > 20


def test_empty_scene_builds():
    sc = S.make_soup(16, 16, 1, 0.1, seed=3)
    R.build_lists(sc)
    assert len(sc.grid_list) >= 1
