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


# ---- against the INDEPENDENT builder oracle (oracle/rt_oracle_builders.c) -----------------------------------------------------
# That file restates trianglelist.cpp serially in its own code (keys + sort, like the reference) and shares nothing with
# rt_build_shared.h.  Still "parity unpinned" against the reference itself (trianglelist.cpp cannot be compiled here), but no
# longer a comparison of the product with its own twin.

def _builder_cases():
    rng = np.random.Generator(np.random.PCG64(4))
    cases = [
        S.make_soup(72, 56, 1200, 0.12, seed=77),
        S.make_soup(200, 150, 2500, 0.08, seed=20),                     # several tiles, ragged edges
        S.make_soup(64, 48, 300, 1.4, seed=8),                          # triangles larger than the image, many off-screen vertices
        S.make_soup(97, 61, 900, 0.3, seed=9, depth=(0.2, 6.0)),        # near triangles: huge projections, some straddle the image border
    ]
    deg = S.make_soup(56, 40, 900, 0.12, seed=18)                      # zero-area and axis-parallel triangles
    v = deg.vertex.reshape(-1, 3, 4)
    v[::7, 1] = v[::7, 0]; v[::7, 2] = v[::7, 0]
    v[::11, 2] = v[::11, 1]
    v[::5, 1, 1] = v[::5, 0, 1]                                         # horizontal edge ab (slope division by zero, :143-150)
    v[::3, 2, 0] = v[::3, 1, 0]                                         # vertical edge bc
    cases.append(deg)
    snap = S.make_soup(80, 60, 600, 0.2, seed=10)                       # vertices exactly on pixel corners / split planes
    snap.vertex[:, :3] = np.round(snap.vertex[:, :3] * 16) / 16 + np.float32(0)  # (+0 turns -0.0 into +0.0: where equal keys land is the sort's business, in the reference too)
    cases.append(snap)
    del rng
    return cases


@pytest.mark.parametrize("idx", range(6))
def test_host_builders_equal_independent_oracle(idx):
    sc = _builder_cases()[idx]
    R.build_lists(sc, threads=4)
    ostart, oend, olist = O.oracle_camera_list(sc)
    assert np.array_equal(sc.cam_start, ostart), "camera Start (incl. the neighbour aliasing of :580-613)"
    assert np.array_equal(sc.cam_end, oend), "camera End"
    assert np.array_equal(sc.cam_list, olist), "camera list (membership, ascending order, de-duplicated storage)"
    obox, ogstart, oglist = O.oracle_scene_grid(sc)
    assert sc.box_min.tobytes() == obox.tobytes(), "split planes"
    assert np.array_equal(sc.grid_start, ogstart), "grid Start"
    assert np.array_equal(sc.grid_list, oglist), "grid list"


def test_builder_oracle_function_level():
    """Projection and box/triangle overlap, spot-checked against first principles (not against the product)."""
    L = O.oracle()
    fp = C.POINTER(C.c_float)
    sc = S.make_soup(64, 48, 10, 0.1, seed=1)
    # a point on the ray through pixel corner (x, y) projects to (x, y) (trianglelist.cpp:74-90)
    for x, y, k in [(0, 0, 1.0), (10, 7, 2.5), (63.5, 47.25, 0.75)]:
        v = (sc.eye[:3] + np.float32(k) * (sc.eye_to_top_left[:3] + sc.left_to_right[:3] * np.float32(x) + sc.top_to_bottom[:3] * np.float32(y))).astype(np.float32)
        out = np.zeros(2, np.float32)
        L.rt_oracle_camera_position(O._f3p(sc.eye), O._f3p(sc.eye_to_top_left), O._f3p(sc.left_to_right), O._f3p(sc.top_to_bottom),
                                    sc.pixel_size_inv, np.ascontiguousarray(v).ctypes.data_as(fp), out.ctypes.data_as(fp))
        assert np.allclose(out, (x, y), atol=1e-3)
    f = lambda *a: np.array(a, np.float32).ctypes.data_as(fp)
    lo, hi = np.array([0, 0, 0], np.float32), np.array([1, 1, 1], np.float32)
    inside = L.rt_oracle_box_meets_triangle(lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), f(0.2, 0.2, 0.5), f(0.8, 0.2, 0.5), f(0.5, 0.8, 0.5))
    through = L.rt_oracle_box_meets_triangle(lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), f(-5, -5, 0.5), f(5, -5, 0.5), f(0, 9, 0.5))
    apart = L.rt_oracle_box_meets_triangle(lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), f(2, 2, 2), f(3, 2, 2), f(2, 3, 2))
    corner = L.rt_oracle_box_meets_triangle(lo.ctypes.data_as(fp), hi.ctypes.data_as(fp), f(1.4, -0.2, 0.5), f(1.4, 0.4, 0.5), f(0.8, -0.2, 0.5))
    assert inside and through and not apart
    assert corner  # its bounding box overlaps and the clipped polygon is not empty: a diagonal cut through the corner region
