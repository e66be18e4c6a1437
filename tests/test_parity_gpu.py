"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  * the golden fixtures minted from the reference kernel (bit-exact u16 planes),
  * the CPU oracle on freshly seeded scenes and on sampled pixel rows of a full-size frame,
  * size-independent properties at BASELINE sizes (tile-partition invariance, plane accumulation, determinism).
Bar: bit-exact.  The only floating-point latitude the design has (double pow/exp2 on the device vs glibc pow,
DESIGN.md) is exercised by `spot_finite_range` and still expected to match exactly on these fixtures."""
import os

import numpy as np
import pytest

import oracle_lib as O
from conftest import golden_names, load_golden_scene
from opencl_render_amd import raytrace as R, scene as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def need_gpu(hip_lib):
    if hip_lib.rtHipDeviceCount() < 1:
        pytest.fail("no HIP device: the GPU parity tests cannot run (and the product has no CPU fallback)")


def assert_planes(got, want, what):
    for ch, g, w in zip("RGB", got, want):
        g = np.asarray(g).reshape(np.asarray(w).shape)
        bad = int((g != w).sum())
        assert bad == 0, f"{what}: plane {ch} differs in {bad}/{g.size} pixels, max |d|={int(np.abs(g.astype(int) - w.astype(int)).max())}"


@pytest.mark.parametrize("name", golden_names())
def test_dropin_raytraceall_matches_golden(name):
    sc, want = load_golden_scene(name)
    ok, r, g, b = R.raytrace_all(1, sc)
    assert ok, R.last_error()
    assert_planes((r, g, b), want, name)


@pytest.mark.parametrize("name", ["mixed_materials_textured", "mirror_hall", "odd_size_multi_tile"])
def test_resident_layer_matches_golden_and_counts_work(name):
    sc, want = load_golden_scene(name)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        got = rs.readback()
        assert_planes(got, want, name)
        # the counted variant computes the same image and the same work counters as the oracle
        stats = rs.render_counted()
        got2 = rs.readback([np.zeros(sc.pixels, np.uint16) for _ in range(3)])
        assert_planes(got2, want, name + " (counted)")
        _, ostats = O.oracle_render(sc, threads=os.cpu_count() or 1, with_stats=True)
        assert stats == ostats
        ms, n = rs.kernel_time_ms()
        assert n == 1 and ms > 0
    finally:
        rs.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_tile_partition_is_invisible(world):
    """round-robin 128x128 tiles, each 'rank' renders only its slice of the camera lists; the sum of the
    ranks' planes is the full frame (the RNG seed uses the global pixel id)."""
    sc, want = load_golden_scene("odd_size_multi_tile")
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    for rank in range(world):
        tiles = R.tiles_of_rank(sc.width, sc.height, rank, world)
        if len(tiles) == 0:
            continue
        rs = R.ResidentScene(sc, 0, tiles)
        try:
            rs.render()
            rs.readback(planes)
        finally:
            rs.close()
    assert_planes(planes, want, f"world={world}")


def test_readback_accumulates_with_saturation():
    sc, want = load_golden_scene("degenerate_and_outside")
    planes = [np.full(sc.pixels, 60000, np.uint16) for _ in range(3)]
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        rs.readback(planes)
    finally:
        rs.close()
    for p, w in zip(planes, want):
        assert np.array_equal(p.reshape(w.shape), np.minimum(60000 + w.astype(np.int64), 65535).astype(np.uint16))


@pytest.mark.parametrize("seed", [5, 6])
def test_fresh_scenes_match_oracle(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    mats = [dict(color=rng.integers(0, 256, (6, 6, 3)), reflection=tuple(rng.integers(0, 160, 3)),
                 transparency=tuple(rng.integers(0, 160, 3)), bump=rng.integers(0, 256, (8, 8, 3)),
                 luminance=tuple(rng.integers(0, 40, 3))) for _ in range(4)]
    lights = [dict(type=int(t), pos=tuple(rng.uniform(-1, 1, 3) + [0, 0, 2]), dir=tuple(rng.uniform(-1, 1, 3)),
                   col=tuple(rng.uniform(0, 1, 3)), radius=float(rng.uniform(0, 1)), half_att=float(rng.choice([np.inf, 2.5])))
              for t in rng.integers(0, 10, 4)]
    sc = S.make_soup(300, 200, 6000, 0.15, seed=seed, samples=3, materials=mats, lights=lights, random_uv=True, smooth_normals=True)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    got = R.render_resident(sc, 0)
    assert_planes(got, want, f"fresh seed {seed}")


@pytest.mark.parametrize("tris,edge", [(1, 0.001), (1, 5.0), (3, 2.0)])
def test_scenes_of_one_to_three_triangles(tris, edge):
    """The smallest inputs the ABI can carry: one speck, one triangle larger than the view, three of them (every grid plane of an
    axis is one of <= 9 vertex coordinates: 256 cells share a handful of distinct planes)."""
    sc = S.make_soup(130, 70, tris, edge, seed=3, samples=1)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=4)
    assert_planes(R.render_resident(sc, 0), want, f"{tris} triangle(s), resident layer")
    ok, r, g, b = R.raytrace_all(1, sc)
    assert ok
    assert_planes((r, g, b), want, f"{tris} triangle(s), drop-in")


def test_scene_without_triangles():
    """No geometry at all (empty vertex / index arrays, empty lists built by the library): black planes from both layers, no error."""
    sc = S.make_soup(70, 50, 1, 0.01, seed=3, samples=2)
    sc.vertex = np.zeros((0, 4), np.float32); sc.tri_index = np.zeros((0, 4), np.int32); sc.tri_material = np.zeros(0, np.int32)
    sc.tri_uv = np.zeros((0, 2), np.float32); sc.tri_normal = np.zeros((0, 4), np.float32)
    R.build_lists(sc)
    assert sc.cam_list.size == 0 and sc.grid_list.size == 0
    want = O.oracle_render(sc, threads=2)
    assert int(np.max(want[0])) == 0
    assert_planes(R.render_resident(sc, 0), want, "empty scene, resident layer")
    ok, r, g, b = R.raytrace_all(1, sc)
    assert ok
    assert_planes((r, g, b), want, "empty scene, drop-in")


def test_crowded_cells_take_the_flat_list_and_the_in_place_loop():
    """The quantile grid keeps cells sparse (a soup has <= 4 candidates per cell), so crowded cells are made: stacks of identical
    and of slightly shifted triangles.  Cells with 2-14 candidates put their further candidates on the trace kernel's per-wave
    second list (which overflows here: 128 entries), cells with 15 or more read their exact count from the first further record
    and loop in place (rt_device.h, pairRec).  Equal t among coincident candidates: the earliest list entry wins, as in the
    reference's running maximum."""
    sc = S.make_soup(320, 200, 3000, 0.05, seed=31, samples=2)
    rng = np.random.default_rng(31)
    verts, tris = [sc.vertex], [sc.tri_index]
    base = sc.vertex_count
    for stack, shift in ((40, 0.0), (14, 0.0), (9, 1e-4), (20, 3e-5)):
        t = int(rng.integers(0, sc.triangle_count))
        v = sc.vertex[sc.tri_index[t, :3]].copy()
        v[:, :3] *= 6.0; v[:, :3] -= v[:, :3].mean(axis=0) * 0.8  # a big copy somewhere in view
        for k in range(stack):
            w = v.copy(); w[:, :3] += np.float32(shift * k)
            verts.append(w); tris.append(np.array([[base, base + 1, base + 2, 0]], np.int32)); base += 3
    n_new = sum(len(t) for t in tris[1:])
    sc.vertex = np.concatenate(verts).astype(np.float32)
    sc.tri_index = np.concatenate(tris).astype(np.int32)
    sc.tri_material = np.concatenate([sc.tri_material, np.zeros(n_new, np.int32)])
    sc.tri_uv = np.concatenate([sc.tri_uv, np.zeros((3 * n_new, 2), np.float32)])
    sc.tri_normal = np.concatenate([sc.tri_normal, np.tile(sc.tri_normal[:3], (n_new, 1))])
    R.build_lists(sc)
    per_cell = np.diff(sc.grid_start.astype(np.int64))
    assert per_cell.max() >= 40 and ((per_cell > 1) & (per_cell < 15)).sum() > 300 and (per_cell >= 15).sum() > 300
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    assert_planes(R.render_resident(sc, 0), want, "crowded cells")


@pytest.mark.parametrize("name", ["mirror_hall", "mixed_materials_textured", "sparse_many_samples"])
def test_megakernel_pipeline_matches_golden(name):
    """The single-launch variant (pipeline 0) ships too: same planes."""
    sc, want = load_golden_scene(name)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.set_pipeline(R.PIPELINE_MEGAKERNEL)
        rs.render()
        assert_planes(rs.readback(), want, name + " (megakernel)")
        rs.set_pipeline(R.PIPELINE_WAVEFRONT)
        rs.render()
        assert_planes(rs.readback([np.zeros(sc.pixels, np.uint16) for _ in range(3)]), want, name + " (wavefront)")
    finally:
        rs.close()


def test_sample_batches_accumulate_in_order(monkeypatch):
    """S=5 with a path-state budget that only fits 1 or 2 samples per batch: the per-sample truncated, saturating
    accumulate (raytrace_opencl.c:726-741) must give the same planes however the samples are batched."""
    sc, want = load_golden_scene("sparse_many_samples")  # 80x60, S=5
    for mb in ("20", "40"):  # ~1 and ~2 samples per batch for one 128x128 tile
        monkeypatch.setenv("RT_WF_STATE_MB", mb)
        rs = R.ResidentScene(sc, 0)
        try:
            rs.stage_timing(True)
            rs.render()
            got = rs.readback()
            _, rounds = rs.stage_times_ms()
        finally:
            rs.close()
        assert_planes(got, want, f"state budget {mb} MB")
    monkeypatch.delenv("RT_WF_STATE_MB")


def test_determinism_two_renders_identical():
    sc, _ = load_golden_scene("mirror_hall")
    a = R.render_resident(sc, 0)
    b = R.render_resident(sc, 0)
    assert_planes(a, b, "repeat")


@pytest.fixture(scope="module")
def full_size_scene():
    """BASELINE config 2 geometry: 1920x1080, 100k-triangle soup (Lambert + one distant light here, so the grid path
    runs too)."""
    sc = S.make_soup(1920, 1080, 100_000, 0.01, seed=12345, samples=1)
    R.build_lists(sc)
    return sc


def test_full_size_sampled_rows_match_oracle(full_size_scene):
    sc = full_size_scene
    got = R.render_resident(sc, 0)
    rows = [0, 1, 269, 540, 811, 1079]
    for y in rows:
        want = O.oracle_render(sc, threads=os.cpu_count() or 1, first_pixel=y * sc.width, pixel_count=sc.width)
        for ch, g, w in zip("RGB", got, want):
            assert np.array_equal(g[y], w[y]), f"row {y} plane {ch}: {(g[y] != w[y]).sum()} pixels differ"
    assert (got[0] > 0).mean() > 0.05


def test_full_size_partition_and_primary_only_properties(full_size_scene):
    sc = full_size_scene
    full = R.render_resident(sc, 0)
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    for rank in range(8):  # the 8-GPU deal, rehearsed on one device
        rs = R.ResidentScene(sc, 0, R.tiles_of_rank(sc.width, sc.height, rank, 8))
        try:
            rs.render()
            rs.readback(planes)
        finally:
            rs.close()
    assert_planes(planes, full, "8-way tile deal at 1080p")
    # primary-only material: out = luminance => every lit pixel is exactly 65535, rest 0
    import copy
    po = copy.copy(sc)
    po.mat_size, po.mat_start, po.textures = S.pack_materials([S.primary_only_material(0)])
    po.tri_material = np.zeros(sc.triangle_count, np.int32)
    empty = S.pack_lights([])
    po.light_type, po.light_pos, po.light_dir, po.light_col, po.light_radius, po.light_half_att = empty
    r, g, b = R.render_resident(po, 0)
    assert set(np.unique(r).tolist()) <= {0, 65535} and np.array_equal(r, g) and np.array_equal(g, b)
    assert np.array_equal(r > 0, full[0] > 0)  # same jitter, same nearest hits: the lit mask is identical


@pytest.mark.parametrize("env", [
    {"RT_WF_SEG": "16,16,16,16,16", "RT_WF_SEG_RAYS": "1,1,1,1"},   # every ray cut into segments of ~16 cell visits, whatever the round size
    {"RT_WF_SEG": "40,24,12,8,8"},                            # finer still for small rounds
    {"RT_WF_SEG": "4096,4096,4096,4096,4096"},                  # never cut
    {"RT_WF_APPEND_RAYS": "0"},                            # every round is an ordered one (planned by the logic kernel, counting sort)
    {"RT_WF_APPEND_RAYS": "4000000000", "RT_WF_ORDERED_FIRST": "0", "RT_WF_SEG": "24,24,24,24,24", "RT_WF_SEG_RAYS": "1,1,1,1"},  # no round is ordered: the trace kernel plans and cuts every ray
    {"RT_WF_ORDERED_FIRST": "0", "RT_WF_GROUP_RAYS": "16"},    # ... in workgroups of 16 rays
    {"RT_WF_LOOKAHEAD": "0"},                              # one ray in flight per path
    {"RT_WF_GROUPS": "3"},                                 # three concurrent tile groups per instance
    {"RT_WF_GROUPS": "2", "RT_WF_SEG": "16,16,16,16,16", "RT_WF_SEG_RAYS": "1,1,1,1", "RT_WF_LOOKAHEAD": "0"},
    {"RT_WF_SLICE_RAYS": "0"},                              # every round spreads its entries over all 256 queue slices per kind
    {"RT_WF_SLICE_RAYS": "4000000000", "RT_WF_SMALL_SLICES": "1", "RT_WF_APPEND_RAYS": "4000000000", "RT_WF_ORDERED_FIRST": "0"},  # one slice per kind from round 1 on, nothing ordered
    {"RT_WF_SLICE_RAYS": "20000", "RT_WF_SMALL_SLICES": "4", "RT_WF_APPEND_RAYS": "0", "RT_WF_SEG": "24,24,24,24,24", "RT_WF_SEG_RAYS": "1,1,1,1"},  # slices merge mid-frame, every round ordered and cut
    {"RT_WF_BLOCKING": "1"},                                # every batch of every frame watched
    {"RT_WF_FAST_QUOTIENT": "0"},                          # every wave divides the long way (the default picks per wave: test_kat_gpu.py)
])
def test_pipeline_modes_are_invisible_in_the_planes(monkeypatch, env):
    """Ray segmentation (the DDA state at a ray parameter is computed without walking, segments are traced independently and
    combined by the lowest segment with a hit), the look-ahead ray and the tile groups only change WHEN cells are visited.
    Golden scenes cover mirrors (rays through the whole grid), transparency chains (finite shadow rays, never cut) and
    wrapping UVs; the big soup has walks of several hundred cells."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for name in ("mirror_hall", "mixed_materials_textured", "all_light_types", "degenerate_and_outside", "spot_finite_range"):
        sc, want = load_golden_scene(name)
        assert_planes(R.render_resident(sc, 0), want, f"{name} with {env}")
    sc = S.make_soup(640, 360, 60_000, 0.012, seed=77, samples=2)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()   # watched frame (guessed layouts)
        assert_planes(rs.readback(), want, f"640x360 soup with {env}, watched frame")
        rs.render()   # planned frame (layouts from the plan)
        assert not rs.finish()
        assert_planes(rs.readback(), want, f"640x360 soup with {env}, planned frame")
    finally:
        rs.close()


def test_dropin_all_gpus_mode_threads_and_tile_deal(monkeypatch):
    """RaytraceAll's extra id "all N GPUs (tiled)" (SURVEY 8b): one scene and one host thread per device, tiles dealt round
    robin, the planes accumulated from every device's tile buffer.  RT_HIP_VIRTUAL_DEVICES deals the tiles over k instances
    on the devices that are there, so the path runs on a one-GPU box."""
    n = R.lib().rtHipDeviceCount()
    for k in ("3", "5"):
        monkeypatch.setenv("RT_HIP_VIRTUAL_DEVICES", k)
        sc, want = load_golden_scene("odd_size_multi_tile")
        ok, r, g, b = R.raytrace_all(n + 1, sc)
        assert ok, R.last_error()
        assert_planes((r, g, b), want, f"odd_size_multi_tile over {k} instances")
    sc = S.make_soup(640, 360, 60_000, 0.012, seed=77, samples=2)  # 5 x 3 tiles
    R.build_lists(sc)
    ok, r, g, b = R.raytrace_all(n + 1, sc)
    assert ok, R.last_error()
    assert_planes((r, g, b), O.oracle_render(sc, threads=os.cpu_count() or 1), "640x360 soup over 5 instances")


@pytest.mark.parametrize("planner", ["logic", "trace"])
def test_every_ray_cut_into_short_segments(monkeypatch, planner):
    """Every ray of a full-coverage 640x360 frame is cut into segments of ~8 cell visits.
    planner = logic: every round is an ordered one (the logic kernel plans and cuts); the further segments of the first rounds
    (several million) do not fit region B of the entry arrays (1 x capacity here), so hundreds of waves find it full while others
    still fit.  A reservation that does not fit must leave its rays whole and must not disturb anybody else's slots (the add is
    never undone, readers clamp the count, the straddling range is marked empty: wf_logic_kernel).
    planner = trace: no round is ordered (the trace kernel's workgroups plan and cut, also the dense first round): a workgroup's
    128 rays would make far more than its 256 lanes' worth of segments, so every workgroup cuts coarser (wf_trace_kernel<false>)."""
    monkeypatch.setenv("RT_WF_SEG", "8,8,8,8,8")
    monkeypatch.setenv("RT_WF_SEG_RAYS", "1,1,1,1")
    if planner == "logic":
        monkeypatch.setenv("RT_WF_APPEND_RAYS", "0")
        monkeypatch.setenv("RT_WF_EXTRA_FACTOR", "1")
    else:
        monkeypatch.setenv("RT_WF_APPEND_RAYS", "4000000000")
        monkeypatch.setenv("RT_WF_ORDERED_FIRST", "0")
        monkeypatch.setenv("RT_WF_GROUP_RAYS", "128")
    sc = S.make_soup(640, 360, 40_000, 0.06, seed=31, samples=1)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        got = rs.readback()
        assert (np.asarray(got[0]) > 0).mean() > 0.85  # nearly every pixel is a path
        assert_planes(got, want, f"every ray cut at 8 visits, planned by the {planner} kernel, watched frame")
        rs.render()
        assert not rs.finish()
        assert_planes(rs.readback(), want, f"every ray cut at 8 visits, planned by the {planner} kernel, planned frame")
    finally:
        rs.close()


def test_walk_guard_fails_the_frame(monkeypatch):
    """wf_trace_kernel's guard against a walk that never ends (RtWavefront::spinLimit) abandons rays when it trips; the
    frame must then FAIL, not return planes with missing hits.  Forced here with a limit of one walk phase."""
    monkeypatch.setenv("RT_WF_SPIN_LIMIT", "1")
    sc, _ = load_golden_scene("sparse_many_samples")  # long empty walks
    rs = R.ResidentScene(sc, 0)
    try:
        with pytest.raises(RuntimeError, match="walk guard"):
            rs.render()
            rs.sync()
    finally:
        rs.close()
    monkeypatch.delenv("RT_WF_SPIN_LIMIT")
    assert_planes(R.render_resident(sc, 0), load_golden_scene("sparse_many_samples")[1], "after the guard test")


# ---- BASELINE.json configs at their full sizes -------------------------------------------------------------------------
# Same seeded soups as bench.py's workloads; the oracle (OpenMP, all cores) renders every `step`-th pixel row of
# the frame and those rows must match bit for bit.  Whole frames would take the oracle minutes at these sizes.

def _bench_scene(name):
    import bench
    return bench.make_scene(name, 1)


def _assert_sampled_rows(sc, got, step, what):
    threads = os.cpu_count() or 1
    bad = 0
    rows = range(0, sc.height, step)
    for y in rows:
        want = O.oracle_render(sc, threads=threads, first_pixel=y * sc.width, pixel_count=sc.width)
        for g, w in zip(got, want):
            bad += int((np.asarray(g).reshape(sc.height, sc.width)[y] != w[y]).sum())
    assert bad == 0, f"{what}: {bad} values differ from the oracle on {len(rows)} sampled rows"
    assert (np.asarray(got[0]) > 0).mean() > 0.05, f"{what}: frame is (nearly) black"


def _assert_whole_frame(sc, got, what):
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    assert_planes(got, want, what)
    assert (np.asarray(got[0]) > 0).mean() > 0.05, f"{what}: frame is (nearly) black"


def test_baseline_config1_restated_256x256_10k_whole_frame():
    """BASELINE config 1 as restated by SURVEY 8d (the .c4d file is not in the checkout): 256x256, 10 k triangles, S=1,
    through the drop-in ABI; the WHOLE frame against the oracle."""
    sc = _bench_scene("smoke")
    ok, r, g, b = R.raytrace_all(1, sc)
    assert ok, R.last_error()
    assert_planes((r, g, b), O.oracle_render(sc, threads=os.cpu_count() or 1), "256x256 / 10 k")


def test_baseline_config3_1080p_1m_lambert():
    """BASELINE config 3, the headline workload: 1920x1080, 1 M triangles, Lambert + one distant light; the WHOLE frame against the
    oracle on all cores (one call: ~1 s on the GPU box)."""
    sc = _bench_scene("lambert_1m")
    _assert_whole_frame(sc, R.render_resident(sc, 0), "1920x1080 / 1 M")


def test_baseline_config4_4k_1m_as_8_way_tile_deal():
    """BASELINE config 4: 3840x2160, 1 M triangles, the frame dealt over 8 ranks (128x128 tiles round-robin, each rank with its
    own scene instance and only its slice of the camera lists), rendered here on one device rank after rank; the ranks'
    tile buffers add up to the frame.  The WHOLE frame against the oracle."""
    sc = _bench_scene("lambert_4k")
    planes = [np.zeros(sc.pixels, np.uint16) for _ in range(3)]
    for rank in range(8):
        rs = R.ResidentScene(sc, 0, R.tiles_of_rank(sc.width, sc.height, rank, 8))
        try:
            rs.render()
            rs.readback(planes)
        finally:
            rs.close()
    _assert_whole_frame(sc, planes, "3840x2160 / 1 M as an 8-way tile deal")


def test_baseline_config5_4k_10m():
    """BASELINE config 5 on one GPU: 3840x2160, 10 M triangles, shadow + bounce rays through the wavefront pipeline; the WHOLE
    frame against the oracle (0.04-0.06 M rays/s per core at this size: ~15 s on the GPU box's share of cores)."""
    sc = _bench_scene("lambert_10m_4k")
    _assert_whole_frame(sc, R.render_resident(sc, 0), "3840x2160 / 10 M")


def test_planned_frames_and_a_plan_that_is_too_short(monkeypatch):
    """Frames after a scene's first are issued from a launch plan without any host synchronisation and verified afterwards
    (rtHipFrameFinish).  (1) Planned frames give the same planes as the watched first frame.  (2) With the plan cut to one round
    (RT_WF_PLAN_ROUNDS=1, a test hook) a planned frame is incomplete: finish() must notice, render the frame again and say so."""
    sc, want = load_golden_scene("mirror_hall")  # many rounds
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()                      # discovery frame (watched)
        assert not rs.finish()
        for _ in range(3):
            rs.render()                  # planned frames
        assert_planes(rs.readback(), want, "planned frame")
        assert not rs.finish()
        rs.stage_timing(True)
        rs.render()
        rs.sync()
        _, rounds = rs.stage_times_ms()
        assert rounds >= 4               # mirrors: the plan covers all of them
    finally:
        rs.close()
    monkeypatch.setenv("RT_WF_PLAN_ROUNDS", "1")
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        assert not rs.finish()           # the first frame is always watched
        rs.render()                      # one planned round only: paths are still waiting
        rs.sync(0)
        # sync() has already redone the frame; a second planned frame checked through finish() reports it
        rs.render()
        assert rs.finish() is True
        assert_planes(rs.readback(), want, "frame redone after a too-short plan")
    finally:
        rs.close()


def test_planned_trace_grid_that_is_too_small_is_noticed(monkeypatch):
    """A planned frame sizes its trace launches from the same frame's previous rendering.  Should such a grid ever be smaller
    than the round's entries (forced here: one workgroup), entries go untraced: the kernel raises RT_WF_ERR_GRID and finish()
    renders the frame again with the worst-case grid."""
    monkeypatch.setenv("RT_WF_PLAN_GRID", "tiny")
    sc = S.make_soup(320, 200, 20_000, 0.03, seed=41, samples=1)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        assert not rs.finish()
        rs.render()
        assert rs.finish() is True
        assert_planes(rs.readback(), want, "frame redone after a too-small trace grid")
    finally:
        rs.close()


def test_primary_only_frames_issue_no_empty_rounds():
    """BASELINE config 2 (luminance-only material, no lights): no path ever waits for the grid, so the plan of a later frame is
    primary + one logic round -- no sort / trace launches at all."""
    sc, want = load_golden_scene("primary_only")
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        rs.sync()
        rs.stage_timing(True)
        rs.render()
        rs.sync()
        ms, rounds = rs.stage_times_ms()
        assert rounds == 1 and ms["trace"] == 0.0 and ms["sort"] == 0.0 and ms["primary"] > 0 and ms["logic"] > 0
        assert_planes(rs.readback(), want, "primary_only, planned frame")
    finally:
        rs.close()


def test_dropin_cache_rebuilds_exactly_what_changed():
    """RaytraceAll keeps its last scene resident and rebuilds parts by content hash.  Every variation below must
    give the oracle's planes for the CHANGED inputs: a stale part would show up as the previous picture."""
    import copy
    threads = os.cpu_count() or 1
    base = S.make_soup(256, 192, 8000, 0.05, seed=51, samples=2)
    R.build_lists(base)

    def check(sc, what):
        ok, r, g, b = R.raytrace_all(1, sc)
        assert ok, R.last_error()
        assert_planes((r, g, b), O.oracle_render(sc, threads=threads), what)
        return r

    first = check(base, "first call (scene built)")
    assert np.array_equal(check(base, "second call, nothing changed (scene reused as it is)"), first)
    moved = copy.copy(base)                       # vertices moved: geometry hash changes, and so do both lists
    moved.vertex = base.vertex.copy()
    moved.vertex[0:1500, 0] += np.float32(0.2)  # 500 triangles shifted sideways
    R.build_lists(moved)
    assert not np.array_equal(check(moved, "a vertex moved (scene rebuilt)"), first)
    cam = copy.copy(moved)                        # camera moved: only the camera part is rebuilt
    cam.eye = moved.eye.copy()
    cam.eye[0] += np.float32(0.15)
    R.build_camera_list(cam)
    check(cam, "camera moved (camera lists rebuilt, geometry reused)")
    more = copy.copy(cam)                         # sample count changed: path-state buffers rebuilt
    more.sample_count = 5
    check(more, "sample count changed")
    lit = copy.copy(more)                         # a second light: lights rebuilt, path state re-sized for several lights
    lights = S.pack_lights([dict(type=S.LIGHT_DISTANT, dir=(0.3, -0.8, 0.5)), dict(type=S.LIGHT_SPOT, pos=(0.5, 0.5, 1.0), col=(0.4, 0.5, 0.9), radius=0.1)])
    lit.light_type, lit.light_pos, lit.light_dir, lit.light_col, lit.light_radius, lit.light_half_att = lights
    check(lit, "lights changed")
    dark = copy.copy(lit)                         # materials changed: rebuilt scene
    dark.textures = lit.textures.copy()
    dark.textures[:, :3] //= 2
    check(dark, "texture atlas changed")
    check(base, "back to the first scene")
    R.lib().rtHipCacheClear()


def test_dropin_second_call_on_an_unchanged_1m_scene_is_fast():
    """the second RaytraceAll on an unchanged 1 M-triangle scene hashes its inputs, finds everything
    resident and renders: a few milliseconds of wall time instead of a scene build (the reference rebuilds all of it,
    raytrace.c:330-489)."""
    import time
    sc = _bench_scene("lambert_1m")
    ok, r0, g0, b0 = R.raytrace_all(1, sc)
    assert ok, R.last_error()
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        ok, r, g, b = R.raytrace_all(1, sc)
        walls.append(time.perf_counter() - t0)
        assert ok and np.array_equal(r, r0) and np.array_equal(b, b0)
    print(f"RaytraceAll on the resident 1 M-triangle scene: {[round(1e3 * w, 2) for w in walls]} ms wall (incl. the ctypes wrapper's three plane allocations)")
    assert min(walls) < 0.030, walls
    R.lib().rtHipCacheClear()


@pytest.mark.parametrize("name", ["odd_size_multi_tile", "lambert_distant"])  # 200x150 (rows of whole 16-byte segments) and 64x48
def test_device_detile_store_and_accumulate(name):
    """rtHipDetileStore writes the tiles' pixels into row-major device planes, rtHipDetile adds them with saturation
    (raytrace_opencl.c:729-740); both in 16-byte pieces where the image width allows and pixel by pixel elsewhere."""
    import ctypes as C
    sc, want = load_golden_scene(name)
    L = R.lib()
    rs = R.ResidentScene(sc, 0)
    P = sc.pixels
    planes = L.rtHipDeviceAlloc(0, 6 * P)
    ids_host = np.arange(R.tile_count(sc.width, sc.height), dtype=np.uint32)
    ids = L.rtHipDeviceAlloc(0, ids_host.nbytes)
    assert planes and ids

    def read():
        host = np.zeros(3 * P, np.uint16)
        assert L.rtHipDeviceCopy(0, host.ctypes.data_as(C.c_void_p), planes, 6 * P, 0) == 0
        return host.reshape(3, sc.height, sc.width)

    try:
        rs.render()
        rs.sync()
        ptr, _ = rs.tile_buffer()
        fill = np.full(3 * P, 60000, np.uint16)
        assert L.rtHipDeviceCopy(0, planes, fill.ctypes.data_as(C.c_void_p), 6 * P, 1) == 0
        assert L.rtHipDeviceCopy(0, ids, ids_host.ctypes.data_as(C.c_void_p), ids_host.nbytes, 1) == 0
        args = (0, ptr, ids, len(ids_host), sc.width, sc.height, planes, planes + 2 * P, planes + 4 * P, None)
        assert L.rtHipDetileStore(*args) == 0
        assert_planes(read(), want, name + " (store)")
        assert L.rtHipDetile(*args) == 0 and L.rtHipDetile(*args) == 0
        for g, w in zip(read(), want):
            assert np.array_equal(g, np.minimum(3 * w.astype(np.int64), 65535).astype(np.uint16))
    finally:
        rs.close()
        L.rtHipDeviceFree(0, planes)
        L.rtHipDeviceFree(0, ids)


def test_planned_frames_with_several_sample_batches(monkeypatch):
    """S=5 in batches of one or two samples: the launch plan is the maximum over the first frame's batches, a planned frame issues it
    for every batch, the status kernel adds up what is left over all of them, and the ordered accumulate still sees the samples in
    order (raytrace_opencl.c:726-741)."""
    sc, want = load_golden_scene("sparse_many_samples")  # 80x60, S=5
    for mb in ("20", "40"):
        monkeypatch.setenv("RT_WF_STATE_MB", mb)
        rs = R.ResidentScene(sc, 0)
        try:
            rs.render()
            assert_planes(rs.readback(), want, f"watched frame, budget {mb} MB")
            rs.render()
            rs.render()
            assert not rs.finish()
            assert_planes(rs.readback(), want, f"planned frames, budget {mb} MB")
        finally:
            rs.close()
    monkeypatch.delenv("RT_WF_STATE_MB")


def test_dropin_cache_notices_edits_in_the_last_bytes_of_an_array():
    """the cache's content hash ORed the tail bytes of an array over each other, so an edit confined to the
    last words of an array whose size is not a multiple of 32 bytes could go unnoticed and the PREVIOUS picture was returned.
    Three such edits, each against the oracle's planes of the changed scene: (a) 3 lights, only light 2's type / radius change
    to values whose bits are a subset of light 0's; (b) an odd triangle count, only the last triangle's third index changes;
    (c) a triangle count with T % 8 == 5, only the last material id changes."""
    import copy
    threads = os.cpu_count() or 1
    mats = [dict(color=(255, 255, 255), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(0, 0, 0)),
            dict(color=(40, 90, 255), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(60, 0, 0))]
    lights = [dict(type=S.LIGHT_DISTANT, dir=(0.3, -0.8, 0.5), radius=1.0), dict(type=S.LIGHT_SPOT, pos=(0.4, 0.6, 1.0), col=(0.5, 0.5, 0.9), radius=2.0),
              dict(type=S.LIGHT_SPOT, pos=(-0.5, 0.2, 1.5), col=(0.9, 0.4, 0.3), radius=0.5)]
    base = S.make_soup(200, 150, 2005, 0.12, seed=61, samples=1, materials=mats, lights=lights, material_ids=np.zeros(2005, np.int32))
    assert base.triangle_count % 2 == 1 and base.triangle_count % 8 == 5
    R.build_lists(base)

    def check(sc, what):
        ok, r, g, b = R.raytrace_all(1, sc)
        assert ok, R.last_error()
        assert_planes((r, g, b), O.oracle_render(sc, threads=threads), what)
        return r, g, b

    first = check(base, "first call")
    a = copy.copy(base)
    a.light_radius = base.light_radius.copy(); a.light_radius[2] = 1.0      # bits of 1.0f are those of light 0's radius
    a.light_type = base.light_type.copy(); a.light_type[2] = S.LIGHT_DISTANT   # 3 = light 0's type (1 is a subset of 3's bits)
    got = check(a, "(a) only light 2's type and radius changed")
    assert any(not np.array_equal(x, y) for x, y in zip(got, first))
    b = copy.copy(a)
    b.tri_index = a.tri_index.copy()
    last = b.triangle_count - 1
    b.tri_index[last, 2] = b.tri_index[0, 2]                               # the last triangle now ends at triangle 0's vertex
    R.build_lists(b)
    check(b, "(b) only the last triangle's third index changed")
    c = copy.copy(b)
    c.tri_material = b.tri_material.copy(); c.tri_material[last] = 1       # T % 8 == 5: the id sits in a 20-byte tail
    check(c, "(c) only the last material id changed")
    R.lib().rtHipCacheClear()


def test_more_lights_than_the_logic_kernel_keeps_in_lds():
    """wf_logic_kernel stages the first 64 lights in LDS and reads the others from global memory; 70 lights of
    several types put both sides of that split on every hit's light loop (the oracle's planes, bit for bit)."""
    rng = np.random.Generator(np.random.PCG64(9))
    lights = []
    for i in range(70):
        t = int(rng.choice([S.LIGHT_DISTANT, S.LIGHT_SPOT, S.LIGHT_OMNI, S.LIGHT_AREA, S.LIGHT_PARALLEL]))
        lights.append(dict(type=t, pos=tuple(rng.uniform(-1, 1, 3) + [0, 0, 1.5]), dir=tuple(rng.uniform(-1, 1, 3)),
                           col=tuple(rng.uniform(0, 0.08, 3)), radius=float(rng.uniform(0.05, 1)), half_att=float(rng.choice([np.inf, 3.0]))))
    sc = S.make_soup(96, 64, 1500, 0.2, seed=19, samples=1, lights=lights)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    got = R.render_resident(sc, 0)
    assert_planes(got, want, "70 lights")
    # the lights past the 64th matter: without them the picture differs
    sc64 = S.make_soup(96, 64, 1500, 0.2, seed=19, samples=1, lights=lights[:64])
    R.build_lists(sc64)
    assert not np.array_equal(O.oracle_render(sc64, threads=os.cpu_count() or 1)[0], want[0])


# ---- the plugin's own operating point -----------------------------------------------------------------------------------------
# The dialog's defaults are 1024 x 768 at 100 samples per pixel (reference render.cpp:176-182): many sample batches per frame, the
# per-sample truncated saturating accumulate (raytrace_opencl.c:726-741) at full depth, the further batches of a watched frame issued
# from the first batch's launch plan.  Sampled rows against the OpenMP oracle (the whole frame is 78 M primary samples).

def test_dialog_defaults_1024x768_100_samples():
    sc = S.make_soup(1024, 768, 100_000, 0.02, seed=2024, samples=100, name="dialog defaults")
    R.build_lists(sc)
    ok, r, g, b = R.raytrace_all(1, sc)      # the first (watched) frame of a scene, as the plugin renders it
    assert ok, R.last_error()
    _assert_sampled_rows(sc, (r, g, b), 24, "1024x768, S=100, drop-in")
    ok, r2, g2, b2 = R.raytrace_all(1, sc)   # the resident scene again: every batch from the plan
    assert ok and np.array_equal(r, r2) and np.array_equal(g, g2) and np.array_equal(b, b2)
    assert len(np.unique(r)) > 2000           # 100 samples: a smooth picture, not a few levels
    R.lib().rtHipCacheClear()


def test_1080p_16_samples_in_several_batches(monkeypatch):
    """1920x1080 at S=16 with a path-state budget that forces four batches of four samples at a real size."""
    monkeypatch.setenv("RT_WF_STATE_MB", "16000")
    sc = S.make_soup(1920, 1080, 300_000, 0.008, seed=77, samples=16, name="1080p S=16")
    R.build_lists(sc)
    rs = R.ResidentScene(sc, 0)
    try:
        rs.render()
        got = rs.readback()
        _assert_whole_frame(sc, got, "1920x1080, S=16, watched frame with planned further batches")
        rs.render()
        assert not rs.finish()
        again = rs.readback()
        assert all(np.array_equal(a, b) for a, b in zip(got, again))
    finally:
        rs.close()


def test_dropin_cache_keeps_the_parts_an_edit_does_not_touch():
    """SURVEY 8f row 2 to the letter: geometry, grid, materials, lights and camera lists are separate parts of RaytraceAll's
    resident scene.  A changed texel re-bakes the materials and leaves triangle records, shading rows and the grid's pair records
    where they are (same device addresses, no free + upload); a moved light touches neither; every picture is the oracle's."""
    import copy
    import ctypes as C
    threads = os.cpu_count() or 1
    L = R.lib()
    mats = [dict(color=np.random.default_rng(4).integers(0, 256, (16, 16, 3)), reflection=(0, 0, 0), transparency=(0, 0, 0), bump=(0, 0, 0), luminance=(0, 0, 0))]
    base = S.make_soup(256, 192, 9000, 0.05, seed=71, samples=2, materials=mats, random_uv=True)
    R.build_lists(base)

    def check(sc, what):
        ok, r, g, b = R.raytrace_all(1, sc)
        assert ok, R.last_error()
        assert_planes((r, g, b), O.oracle_render(sc, threads=threads), what)
        ptr = (C.c_void_p * 6)()
        assert L.rtHipTestCachePointers(ptr) == 0
        return r, [p for p in ptr]

    first, p0 = check(base, "first call")
    texel = copy.copy(base)
    texel.textures = base.textures.copy()
    texel.textures[:256, :3] = 255 - texel.textures[:256, :3]   # (the colour image's texels: the one-texel channels behind it stay black)
    pic, p1 = check(texel, "texture atlas changed")
    assert not np.array_equal(pic, first)
    assert p1[0] == p0[0] and p1[1] == p0[1] and p1[2] == p0[2], "geometry / grid were rebuilt for a texel edit"
    light = copy.copy(texel)
    light.light_dir = texel.light_dir.copy()
    light.light_dir[0, :3] = (-0.4, -0.7, 0.3)
    pic2, p2 = check(light, "light moved")
    assert not np.array_equal(pic2, pic) and p2[:5] == p1[:5], "a moved light rebuilt geometry, grid or materials"
    grid = copy.copy(light)                       # same triangles, another grid (here: the lists of a coarser neighbour scene are not
    grid.vertex = light.vertex.copy()             # available, so the geometry moves too: everything but the materials is rebuilt)
    grid.vertex[:300, 1] += np.float32(0.1)
    R.build_lists(grid)
    pic3, p3 = check(grid, "vertices moved")
    assert p3[3] == p2[3] and p3[4] == p2[4], "materials were rebuilt for a geometry edit"
    check(base, "back to the first scene")
    L.rtHipCacheClear()


def test_all_gpus_mode_builds_the_scene_once(monkeypatch):
    """RaytraceAll's all-GPUs id builds the scene on the first instance and copies geometry, grid, materials and lights to the others
    device to device ('upload once via root then broadcast'); on a one-GPU box the instances share the device
    (virtual_devices).  Four instances of the 1 M-triangle scene: sampled rows against the oracle, and a first call that costs
    little more than one instance's (it used to upload and reshape the scene once per instance)."""
    import time
    n = R.lib().rtHipDeviceCount()
    sc = _bench_scene("lambert_1m")
    one = four = 1e9
    for rep in range(2):  # (the first call of either kind also pays the driver's first mapping of that much memory: the better of two is compared)
        R.lib().rtHipCacheClear()
        monkeypatch.delenv("RT_HIP_VIRTUAL_DEVICES", raising=False)
        t0 = time.perf_counter()
        ok, r1, g1, b1 = R.raytrace_all(1, sc)
        one = min(one, time.perf_counter() - t0)
        assert ok, R.last_error()
        R.lib().rtHipCacheClear()
        monkeypatch.setenv("RT_HIP_VIRTUAL_DEVICES", "4")
        t0 = time.perf_counter()
        ok, r, g, b = R.raytrace_all(n + 1, sc)
        four = min(four, time.perf_counter() - t0)
        assert ok, R.last_error()
        assert np.array_equal(r, r1) and np.array_equal(g, g1) and np.array_equal(b, b1)
    _assert_whole_frame(sc, (r, g, b), "1920x1080 / 1 M over 4 instances")
    print(f"first RaytraceAll on the 1 M-triangle scene: one instance {1e3 * one:.1f} ms, four instances {1e3 * four:.1f} ms wall")
    # measured: 17 ms against 35 ms (every instance allocates its own path state, uploads its own camera ranges and renders a watched first
    # frame; the shared parts are copied in under a millisecond each).  Uploading and reshaping the scene per instance was 4 x 24 ms more.
    assert four < 2.5 * one + 0.03, (one, four)
    R.lib().rtHipCacheClear()
