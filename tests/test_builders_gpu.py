"""GPU list builders (rt_build_device.hip) against the host builders (rt_builders.cpp) AND against the independent builder
oracle (oracle/rt_oracle_builders.c, a serial restatement of trianglelist.cpp that shares no code with either).  Camera
lists: the same triangles in the same pixels, every pixel's entries ascending, and the same storage sharing between equal
neighbouring lists (the reference's de-duplication, trianglelist.cpp:580-613) -- so Start, End and the list are compared as
arrays.  Parity of the builders against the reference itself stays unpinned (trianglelist.cpp cannot be compiled here)."""
import copy

import numpy as np
import pytest

from opencl_render_amd import raytrace as R, scene as S

pytestmark = pytest.mark.gpu


def per_pixel_lists(sc):
    return [sc.cam_list[a:b] for a, b in zip(sc.cam_start.tolist(), sc.cam_end.tolist())]


def assert_same_lists(host, dev, label):
    assert len(host.cam_start) == len(dev.cam_start) == host.pixels
    n_host = host.cam_end.astype(np.int64) - host.cam_start.astype(np.int64)
    n_dev = dev.cam_end.astype(np.int64) - dev.cam_start.astype(np.int64)
    bad = np.nonzero(n_host != n_dev)[0]
    assert bad.size == 0, f"{label}: {bad.size} pixels differ in list length, first {bad[:5]} (host {n_host[bad[:5]]}, device {n_dev[bad[:5]]})"
    # flatten both in pixel order and compare at once
    order_h = np.concatenate([np.arange(a, b) for a, b in zip(host.cam_start.tolist(), host.cam_end.tolist())]) if n_host.sum() else np.zeros(0, np.int64)
    order_d = np.concatenate([np.arange(a, b) for a, b in zip(dev.cam_start.tolist(), dev.cam_end.tolist())]) if n_dev.sum() else np.zeros(0, np.int64)
    assert np.array_equal(host.cam_list[order_h], dev.cam_list[order_d]), f"{label}: list contents differ"


def with_big_triangles(sc, seed):
    """Adds triangles that cover large parts of the image (rectangles far above RT_BIG_RECT pixels), some partly off screen."""
    rng = np.random.default_rng(seed)
    extra = []
    for _ in range(6):
        z = rng.uniform(2.0, 4.0)
        centre = np.array([rng.uniform(-0.3, 0.3) * z, rng.uniform(-0.2, 0.2) * z, z], np.float32)
        pts = centre + rng.uniform(-0.9, 0.9, (3, 3)).astype(np.float32) * np.float32(z * 0.5)
        pts[:, 2] = np.maximum(pts[:, 2], 1.5)
        extra.append(pts.astype(np.float32))
    extra = np.stack(extra)  # [k,3,3]
    k = len(extra)
    out = copy.copy(sc)
    v = np.zeros((3 * k, 4), np.float32)
    v[:, :3] = extra.reshape(-1, 3)
    base = sc.vertex.shape[0]
    out.vertex = np.concatenate([sc.vertex, v])
    idx = np.zeros((k, 4), np.int32)
    idx[:, 0] = base + 3 * np.arange(k); idx[:, 1] = idx[:, 0] + 1; idx[:, 2] = idx[:, 0] + 2
    out.tri_index = np.concatenate([sc.tri_index, idx])
    out.tri_material = np.concatenate([sc.tri_material, np.zeros(k, np.int32) + sc.tri_material[0]])
    out.tri_uv = np.concatenate([sc.tri_uv, np.zeros((3 * k, 2), np.float32)])
    n = np.zeros((3 * k, 4), np.float32); n[:, 2] = -1
    out.tri_normal = np.concatenate([sc.tri_normal, n])
    return out


@pytest.mark.parametrize("w,h,tris,edge,seed,big", [
    (300, 200, 6000, 0.15, 5, False),      # many candidates per pixel
    (640, 360, 60_000, 0.012, 77, False),  # small triangles
    (333, 177, 3000, 0.4, 9, True),        # odd size, triangles larger than the image, partly off screen
    (1920, 1080, 100_000, 0.01, 12345, True),
])
def test_device_camera_lists_equal_host_lists(w, h, tris, edge, seed, big):
    sc = S.make_soup(w, h, tris, edge, seed=seed, samples=1)
    if big:
        sc = with_big_triangles(sc, seed)
    host, dev = copy.copy(sc), copy.copy(sc)
    R.build_camera_list(host)
    ms = R.build_camera_list_device(dev, 0)
    assert ms > 0
    assert_same_lists(host, dev, f"{w}x{h}, {sc.triangle_count} triangles")
    # with the neighbour de-duplication on the device as well, the three arrays themselves are equal
    assert np.array_equal(host.cam_start, dev.cam_start), "Start (aliasing) differs"
    assert np.array_equal(host.cam_end, dev.cam_end), "End differs"
    assert np.array_equal(host.cam_list, dev.cam_list), "list storage differs"
    if sc.triangle_count <= 10_000:  # the serial oracle sorts 64-bit keys: keep it to the small cases
        import oracle_lib as O
        ostart, oend, olist = O.oracle_camera_list(sc)
        assert np.array_equal(dev.cam_start, ostart) and np.array_equal(dev.cam_end, oend) and np.array_equal(dev.cam_list, olist), \
            "device camera lists differ from the independent oracle"


@pytest.mark.parametrize("w,h,tris,edge,seed,big", [
    (300, 200, 6000, 0.15, 5, False),      # triangles over a few cells each: the one-thread fill, some spill to the workgroup fill
    (640, 360, 60_000, 0.012, 77, False),  # small triangles
    (333, 177, 3000, 0.4, 9, True),        # triangles over thousands of cells: the workgroup fill
    (1920, 1080, 100_000, 0.01, 12345, True),
])
def test_device_grid_equals_host_grid(w, h, tris, edge, seed, big):
    sc = S.make_soup(w, h, tris, edge, seed=seed, samples=1)
    if big:
        sc = with_big_triangles(sc, seed)
    host, dev = copy.copy(sc), copy.copy(sc)
    R.build_scene_grid(host)
    ms = R.build_scene_grid_device(dev, 0)
    assert ms > 0
    assert np.array_equal(host.box_min, dev.box_min), "split planes differ"
    assert len(host.grid_list) == len(dev.grid_list), f"pair count {len(host.grid_list)} vs {len(dev.grid_list)}"
    assert np.array_equal(host.grid_start, dev.grid_start), "cell starts differ"
    assert np.array_equal(host.grid_list, dev.grid_list), "cell lists differ"
    if sc.triangle_count <= 10_000:
        import oracle_lib as O
        obox, ostart, olist = O.oracle_scene_grid(sc)
        assert dev.box_min.tobytes() == obox.tobytes() and np.array_equal(dev.grid_start, ostart) and np.array_equal(dev.grid_list, olist), \
            "device grid differs from the independent oracle"


def test_frame_from_device_built_lists_matches_oracle():
    """The hot path on device-built lists: same planes as the CPU oracle on the host-built lists."""
    import os
    import oracle_lib as O
    sc = S.make_soup(320, 240, 20_000, 0.03, seed=21, samples=2)
    host = copy.copy(sc)
    R.build_lists(host)
    want = O.oracle_render(host, threads=os.cpu_count() or 1)
    dev = copy.copy(sc)
    R.build_camera_list_device(dev, 0)
    R.build_scene_grid_device(dev, 0)
    got = R.render_resident(dev, 0)
    for ch, g, w in zip("RGB", got, want):
        assert np.array_equal(g, w), f"plane {ch}: {(g != w).sum()} values differ"
