#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark (BASELINE.json: Mrays/s + ms/frame at 1920x1080, 1M-triangle scene).

    python bench.py --gpus 1 --steps K --warmup W                       (default: N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame: every rank renders its 128x128 tiles of the frame (one launch of the trace kernel over the
scene resident in its HBM), the tile buffers are gathered to rank 0 (RCCL) and rank 0 scatters them into the three
row-major u16 planes on its device.  Inputs are resident before the timed region; outputs stay on the device for
`value` / `ms_per_step`; `ms_per_frame_with_d2h` is the same loop with the planes copied to pinned host memory every frame,
`ms_per_frame_with_d2h_pipelined` the same with frame i's copy on a second stream under frame i+1's kernels,
`ms_per_frame_watched` a frame nothing is known about (every frame of a second scene instance watched: queue read-backs, guessed layouts).
The same frame is split over more GPUs as N grows => "scaling": "strong".

Workloads (synthetic seeded triangle soups, lists built by the library's host builders):
    lambert_1m   1920x1080, 1M triangles (edge 0.004), white Lambert + 1 distant light, S=1   <- default, BASELINE metric
    primary_100k 1920x1080, 100k triangles (edge 0.01), luminance-only material, no lights (primary rays only)
    lambert_4k   3840x2160, 1M triangles (edge 0.004), as lambert_1m
    lambert_10m_4k  3840x2160, 10M triangles (edge 0.0013), as lambert_1m (BASELINE config 5 on one GPU)
One JSON line on stdout (rank 0); progress goes to stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "lambert_1m": dict(width=1920, height=1080, triangles=1_000_000, edge=0.004, kind="lambert"),
    "primary_100k": dict(width=1920, height=1080, triangles=100_000, edge=0.01, kind="primary"),
    "lambert_4k": dict(width=3840, height=2160, triangles=1_000_000, edge=0.004, kind="lambert"),
    "lambert_10m_4k": dict(width=3840, height=2160, triangles=10_000_000, edge=0.0013, kind="lambert"),
    "smoke": dict(width=256, height=256, triangles=10_000, edge=0.02, kind="lambert"),
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # ... and the measured float4-copy ceiling (report both fractions)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def algorithmic_bytes(stats: dict, lights: int) -> float:
    """SURVEY.md section 8(d) gather-traffic model, no credit for cache reuse:
    B = S*20*P + 68*candidates(primary) + sum over grid rays (8*cells + 68*candidates) + per shaded hit
    (4 matId + 16 index + 48 normals + 48 vertices + 8 bump size) + per texel fetch (8 size + 4 start + 24 uv + 4 texel)
    + boxMin (4112 B) and lights (60 B each) once per frame."""
    return (20.0 * stats["primarySamples"] + 68.0 * stats["primaryCandidates"] + 8.0 * stats["gridCells"] + 68.0 * stats["gridCandidates"]
            + 124.0 * stats["shadedHits"] + 40.0 * stats["texelFetches"] + 4112.0 + 60.0 * lights)


def make_scene(name: str, samples: int):
    from opencl_render_amd import raytrace as R, scene as S
    w = WORKLOADS[name]
    if w["kind"] == "primary":
        sc = S.make_soup(w["width"], w["height"], w["triangles"], w["edge"], seed=12345, samples=samples,
                         materials=[S.primary_only_material(256)], lights=[], random_uv=True, name=name)
    else:
        sc = S.make_soup(w["width"], w["height"], w["triangles"], w["edge"], seed=12345, samples=samples, name=name)
    t0 = time.time()
    R.build_camera_list(sc)
    t1 = time.time()
    R.build_scene_grid(sc)
    t2 = time.time()
    sc.meta.update(t_cam_list_s=t1 - t0, t_grid_s=t2 - t1)
    return sc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="lambert_1m", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from opencl_render_amd import raytrace as R, tiles as T

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if not torch.cuda.is_available() or R.lib().rtHipDeviceCount() < 1:
        sys.exit("bench.py: no HIP device visible -- the HIP path has no CPU fallback")
    # RT_BENCH_REHEARSE=1: functional rehearsal of the N>1 path on ONE GPU -- every rank uses device 0 and the
    # collectives run on gloo through host memory.  Timings of such a run mean nothing; results must still be exact.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- scene: built once on rank 0, shipped to the other ranks over RCCL -----------------------------------------
    sc = None
    if rank == 0:
        t0 = time.time()
        sc = make_scene(args.workload, args.samples)
        log(f"scene {sc.name}: {sc.width}x{sc.height}, T={sc.triangle_count}, cam list {len(sc.cam_list)} (mean K_p "
            f"{sc.sum_candidates() / sc.pixels:.2f}), grid list {len(sc.grid_list)}; host prep {time.time() - t0:.1f}s "
            f"(camera lists {sc.meta['t_cam_list_s']:.2f}s, grid {sc.meta['t_grid_s']:.2f}s)")
    t_dev_cam = t_dev_grid = None
    if rank == 0 and not rehearse:
        # both lists once more on the GPU (rt_build_device.hip; same lists as the host builders', tests/test_builders_gpu.py):
        # reported beside the host builder's time, the frame below uses the host-built lists
        import copy
        probe = copy.copy(sc)
        R.build_camera_list_device(probe, local_rank)  # first call pays the code-object load
        t0 = time.time()
        ms = R.build_camera_list_device(probe, local_rank)
        t_dev_cam = {"device_ms": round(ms, 3), "wall_s_with_transfers": round(time.time() - t0, 4)}
        t0 = time.time()
        ms = R.build_scene_grid_device(probe, local_rank)
        t_dev_grid = {"device_ms": round(ms, 3), "wall_s_with_transfers": round(time.time() - t0, 4)}
        del probe
    if world > 1:
        sc = T.broadcast_scene(sc, rank, device, keep_on_device=not rehearse)  # (receiving ranks: the tensors go to the library as they are)
    W, H, P, S = sc.width, sc.height, sc.pixels, sc.sample_count

    my_tiles = R.tiles_of_rank(W, H, rank, world)
    t0 = time.time()
    rs = R.ResidentScene(sc, local_rank, my_tiles)
    t_upload = time.time() - t0
    if rank == 0:
        log(f"scene resident on the device after {t_upload:.2f}s ({rs.device_bytes() / 1e9:.2f} GB)" if hasattr(rs, "device_bytes") else f"scene resident after {t_upload:.2f}s")
    buf_ptr, buf_bytes = rs.tile_buffer()
    tile_tensor = T.alias_device_bytes(buf_ptr, buf_bytes, device)
    # One real stream for the whole frame: the rays' kernels, the gather and the de-tiling launch must be ordered behind each
    # other ON THE DEVICE (a frame is issued without any host synchronisation).  Stream 0 would mean "the scene's own stream" to
    # the library and the default stream to torch -- two different queues.
    work_stream = torch.cuda.Stream(device)
    stream = work_stream.cuda_stream
    assert stream != 0

    planes = ids_dev = None
    slots = T.max_tiles_per_rank(W, H, world)
    if rank == 0:
        planes = torch.zeros(3 * P * 2, dtype=torch.uint8, device=device)
        # The gathered buffer is [world][slots] tiles: ONE de-tiling launch over all of it.  Slot s of rank r holds tile
        # ids[r][s]; padding slots get the first id past the grid, which the kernel drops at its height test.
        ids = np.full((world, slots), R.tile_count(W, H), np.int32)
        for r in range(world):
            mine = R.tiles_of_rank(W, H, r, world).astype(np.int32)
            ids[r, :len(mine)] = mine
        ids_dev = torch.from_numpy(ids.reshape(-1)).to(device)
    L = R.lib()

    def frame(to_host=None, into=None):
        with torch.cuda.stream(work_stream):
            rs.render(stream)
            gathered = T.gather_tiles(tile_tensor, W, H, rank, world)
            if rank == 0:
                # every pixel of the frame belongs to exactly one tile of the deal: the root writes its planes (no zero fill + add)
                base = (planes if into is None else into).data_ptr()
                rc = L.rtHipDetileStore(local_rank, gathered.data_ptr(), ids_dev.data_ptr(), world * slots, W, H,
                                        base, base + 2 * P, base + 4 * P, stream)
                if rc != 0:
                    raise RuntimeError("rtHipDetileStore: " + R.last_error())
                if to_host is not None:
                    to_host.copy_(planes, non_blocking=True)

    def barrier():
        if world > 1:
            dist.barrier()

    def drained():
        """torch.cuda.synchronize() behind a poll of the frame's stream: the runtime's blocking wait has been seen to return tens of
        milliseconds after the device finished (interrupt-driven wake-ups on a shared host), which a 19 ms timed region cannot absorb."""
        while not work_stream.query():
            pass
        torch.cuda.synchronize()

    # The interpreter's cyclic garbage collector is held off from here to the end of the timed region: a full collection with torch
    # imported walks a few hundred thousand objects -- tens of milliseconds of a host thread that has 19 ms of frames to keep the device
    # fed with (and a collection right in front of the timed frames would let the chip's clocks drop again).
    import gc
    gc.collect()
    gc.disable()
    # ORDER: everything that is not the contract's W + K frames runs FIRST -- the watched-frame figure, the stage-timed and counted
    # passes, the frames with copies to the host -- and the timed region comes last.  The chip takes ~20 ms of continuous work to reach
    # its steady clocks from idle (scripts/x_ramp.py: frames 1-10 of a cold start 0.996 ms, 11-20 0.946, from then on 0.922-0.927): timed
    # right behind the scene build, the same K frames average 2-3 % more than the rate the device sustains.
    # ---- what a frame costs when NOTHING is known about it: every frame watched (the queue read back between round chunks, layouts
    # guessed), as the first frame of a scene or after a camera move is.  `value` is the steady state of a resident scene; this is the
    # other end.  A second instance of the scene, built with the library told to watch every frame.
    ms_watched = None
    if world == 1:
        os.environ["RT_WF_BLOCKING"] = "1"
        try:
            rw = R.ResidentScene(sc, local_rank, my_tiles)
        finally:
            del os.environ["RT_WF_BLOCKING"]
        for _ in range(2):
            rw.render(); rw.sync()
        t3 = time.perf_counter()
        n_watched = max(3, args.steps // 2)
        for _ in range(n_watched):
            rw.render(); rw.sync()
        ms_watched = 1e3 * (time.perf_counter() - t3) / n_watched
        rw.close()

    # ---- un-timed: work counters for the byte model (instrumented kernel variant) ---------------------------------------
    for _ in range(2):  # (the scene's first frame is a watched one and leaves the launch plan behind)
        frame()
    torch.cuda.synchronize()
    stats = rs.render_counted()
    frame()
    torch.cuda.synchronize()
    total_stats = dict(stats)
    if world > 1:
        keys = sorted(stats)
        t = torch.tensor([stats[k] for k in keys], dtype=torch.int64, device="cpu" if rehearse else device)
        dist.all_reduce(t)
        total_stats = {k: int(v) for k, v in zip(keys, t.tolist())}

    # ---- the same frames once more with the three planes copied to (pinned) host memory at the end of every frame, on the
    # frame's stream: ms/frame as SURVEY 8(d) defines it (t_kernel + t_gather + D2H).  `value` stays the device-resident rate.
    ms_with_d2h = None
    if rank == 0:
        host_planes = torch.empty(3 * P * 2, dtype=torch.uint8, pin_memory=True)
    else:
        host_planes = None
    for _ in range(2):
        frame(host_planes)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        frame(host_planes)
    drained()
    barrier()
    ms_with_d2h = 1e3 * (time.perf_counter() - t1) / args.steps
    serial_ref = host_planes.clone() if rank == 0 else None  # what the serial frames left in host memory (the pipelined ones reuse the buffer)
    # ---- and pipelined three deep: frame i's planes travel on a copy stream while frames i+1 and i+2 render into the other two of three
    # plane buffers (events both ways); every frame's planes are complete in pinned host memory when the clock stops.  (Three, not two:
    # the copy is a kernel of the runtime's that shares the chip with the next frame's kernels and at times takes longer than a frame.)
    ms_with_d2h_pipelined = None
    DEPTH = 3
    if rank == 0:
        copy_stream = torch.cuda.Stream(device)
        dev2 = [planes] + [torch.zeros_like(planes) for _ in range(DEPTH - 1)]
        host2 = [host_planes] + [torch.empty(3 * P * 2, dtype=torch.uint8, pin_memory=True) for _ in range(DEPTH - 1)]
        rendered = [torch.cuda.Event() for _ in range(DEPTH)]
        copied = [torch.cuda.Event() for _ in range(DEPTH)]

    def frame_pipelined(i):
        b = i % DEPTH
        if rank == 0 and i >= DEPTH:
            work_stream.wait_event(copied[b])  # the buffer's previous frame has left the device
        frame(into=dev2[b] if rank == 0 else None)
        if rank == 0:
            rendered[b].record(work_stream)
            copy_stream.wait_event(rendered[b])
            with torch.cuda.stream(copy_stream):
                host2[b].copy_(dev2[b], non_blocking=True)
            copied[b].record(copy_stream)

    for i in range(DEPTH):
        frame_pipelined(i)
    torch.cuda.synchronize()
    barrier()
    t2 = time.perf_counter()
    for i in range(args.steps):
        frame_pipelined(i + DEPTH)
    if rank == 0:
        while not copy_stream.query():
            pass
    drained()
    barrier()
    ms_with_d2h_pipelined = 1e3 * (time.perf_counter() - t2) / args.steps
    if rs.finish():
        sys.exit("bench.py: a frame with a copy to the host needed more rounds than its launch plan issued")
    # ---- un-timed: per-stage device time (HIP events around every launch; the launches of a frame no longer run back to back) ----
    # Three passes, per stage the smallest of the three means: what inflates a bracket is one-sided -- the host falling behind the device
    # after a short kernel puts its own launch latency between the start event and the kernel (same box, six runs: trace 0.607-0.649 ms
    # from single passes while the frames themselves took 0.929-0.937 ms every time).
    stage_frames = max(3, min(10, args.steps))
    rs.stage_timing(True)
    stage_ms, rounds = None, 0
    for _ in range(3):
        for _ in range(stage_frames):
            frame()
        torch.cuda.synchronize()
        one, rounds = rs.stage_times_ms()
        stage_ms = {k: v / stage_frames for k, v in one.items()} if stage_ms is None else {k: min(stage_ms[k], v / stage_frames) for k, v in one.items()}
    rs.stage_timing(False)
    # ---- the contract's region: W warm-up frames, then K timed frames between barriers --------------------------------------
    for _ in range(args.warmup):
        frame()
    torch.cuda.synchronize()
    rs.kernel_time_ms()  # drop warm-up launches from the kernel-time average
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    drained()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # (host-side comparison of the pipelined pass's buffers, kept out of the device's way until now)
    if rank == 0 and not all(torch.equal(h, serial_ref) for h in host2):
        sys.exit("bench.py: pipelined frames differ from the serial ones")
    # The timed frames were issued from a launch plan, without looking at the ray queue (rtHipFrameFinish): now that the
    # device is idle, check that every one of them really was complete.  A frame that was not voids the run.
    if rs.finish():
        sys.exit("bench.py: a timed frame needed more rounds than its launch plan issued -- timing is void")
    if rank == 0:
        log(f"timed {args.steps} frames: {1e3 * elapsed / args.steps:.3f} ms/frame")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms, launches = rs.kernel_time_ms()
    rehearsal_bad = 0
    if rank == 0:
        got = planes.cpu().numpy().view(np.uint16).reshape(3, H, W)
        ms_per_step = 1e3 * elapsed / args.steps
        primary_rays = float(P) * S
        total_rays = primary_rays + total_stats["gridRays"]
        b_local = algorithmic_bytes(stats, sc.light_count)  # this rank's frame: its tiles only
        pipeline = os.environ.get("RT_HIP_PIPELINE", "1") != "0"
        # dominant kernel: the grid trace (wf_trace_kernel); its share of the byte model is the grid terms
        b_trace = 8.0 * stats["gridCells"] + 68.0 * stats["gridCandidates"]
        if pipeline and stage_ms["trace"] > 0 and b_trace > 0:
            dom_name, dom_ms, dom_bytes = "wf_trace_kernel (all launches of one frame)", stage_ms["trace"], b_trace
        elif pipeline:
            dom_name, dom_ms, dom_bytes = "wf_primary_kernel", stage_ms["primary"], 20.0 * stats["primarySamples"] + 68.0 * stats["primaryCandidates"]
        else:
            dom_name, dom_ms, dom_bytes = "rt_trace_kernel<false>", kernel_ms, b_local
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes of the dominant kernel per frame from the PMC passes committed under profiles/ (collected by
        # scripts/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, gfx950 x2 read correction);
        # only reported when that profile is of this very workload and N=1
        # `traffic` is NOT measured by this run (counters need rocprofv3): it is the fabric (L2-miss) bytes of the dominant kernel
        # per frame from the counter passes committed under profiles/ -- TCC_EA0_RDREQ x 128 B, every request of this kernel is
        # a 128-byte one (profiles/r03_*) -- and is tagged with the file and the commit it was taken at.  Only given when that
        # profile is of this very workload, S=1, N=1.
        traffic = traffic_source = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r03_trace_fabric_traffic.json")))
            key = dom_name.split(" ")[0]
            rows = [v for k, v in prof["per_frame"].items() if k.split("<")[0] == key]  # (both instantiations of the trace kernel)
            if world == 1 and pipeline and prof.get("workload") == args.workload and args.samples == 1 and rows:
                traffic = int(sum(r["fabric_read_bytes"] + r.get("write_bytes", 0) for r in rows))
                traffic_source = f"profiles/r03_trace_fabric_traffic.json @ {prof.get('commit', '?')} (not measured by this run)"
        except (OSError, ValueError, KeyError):
            traffic = traffic_source = None
        out = {
            "metric": "Mrays/s (primary rays; each also traces its shadow/bounce rays) at ms/frame = ms_per_step",
            "value": round(primary_rays * args.steps / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame_with_d2h": round(ms_with_d2h, 4),
            "ms_per_frame_with_d2h_pipelined": round(ms_with_d2h_pipelined, 4),
            "ms_per_frame_watched": None if ms_watched is None else round(ms_watched, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {sc.triangle_count} random triangles (edge {sc.meta.get('edge')}), "
                                   f"{'luminance-only, primary rays only' if WORKLOADS[args.workload]['kind'] == 'primary' else 'white Lambert + 1 distant light (shadow ray + 1 diffuse bounce)'}"
                                   f", S={S}, seed 12345",
                       "width": W, "height": H, "triangles": sc.triangle_count, "samples": S, "lights": sc.light_count,
                       "parallelism": f"128x128 tiles round-robin over {world} GPU(s), RCCL gather to rank 0"},
            "total_mrays_per_s": round(total_rays * args.steps / elapsed / 1e6, 3),
            "rays_per_frame": {"primary": int(primary_rays), "grid": int(total_stats["gridRays"])},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": dom_name, "kernel_ms": round(dom_ms, 4), "algorithmic_bytes": int(dom_bytes),
                         "frame": {"device_ms": round(kernel_ms, 4), "frames": int(launches), "algorithmic_bytes": int(b_local),
                                   "achieved": round(b_local / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else 0.0,
                                   "frac": round(b_local / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kernel_ms > 0 else 0.0},
                         "stage_ms_per_frame": {k: round(v, 4) for k, v in stage_ms.items()}, "rounds": int(rounds)},
            "work_counters": total_stats,
            "t_upload_s": round(t_upload, 3),
            "t_host_prep_s": {"camera_lists": round(sc.meta.get("t_cam_list_s", 0), 3), "grid": round(sc.meta.get("t_grid_s", 0), 3)},
            "t_device_prep": {"camera_lists": t_dev_cam, "grid": t_dev_grid},
        }
        # ---- CPU baseline + parity gate (rank 0, N=1 only): the oracle on a bounded sample of the same frame ------
        if world > 1 and rehearse:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # checker only
            bad = 0
            rows = list(range(0, H, 97))
            for y in rows:
                want = O.oracle_render(sc, threads=os.cpu_count() or 1, first_pixel=y * W, pixel_count=W)
                for c in range(3):
                    bad += int((want[c][y] != got[c, y]).sum())
            out["parity"] = {"rows_checked": len(rows), "mismatching_values": bad, "bar": "bit-exact u16 planes vs CPU oracle (rehearsal)"}
            rehearsal_bad = bad
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # checker only: never part of the measured path
            rows = list(range(0, H, 1 if P * S <= 2_500_000 else 4))  # ~10 s single thread (the reference's C path is single-threaded)
            log(f"CPU baseline + parity gate: {len(rows)} rows single-threaded, then the whole frame on all cores")
            t0 = time.perf_counter()
            bad = 0
            for y in rows:
                want = O.oracle_render(sc, threads=1, first_pixel=y * W, pixel_count=W)
                for c in range(3):
                    bad += int((want[c][y] != got[c, y]).sum())
            t_cpu = time.perf_counter() - t0
            cores = os.cpu_count() or 1
            t0 = time.perf_counter()
            want_all = O.oracle_render(sc, threads=cores)
            t_all = time.perf_counter() - t0
            rows_checked = len(rows)
            if len(rows) < H:  # the single-thread leg sampled rows: the all-cores frame is the whole-frame gate
                bad += sum(int((np.asarray(want_all[c]).reshape(H, W) != got[c]).sum()) for c in range(3))
                rows_checked = H
            out["cpu_baseline"] = {"value": round(len(rows) * W * S / t_cpu / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                                   "sample": f"{len(rows)} of {H} rows of the same frame ({len(rows) * W * S} primary samples, {t_cpu:.1f}s)",
                                   "all_cores": {"value": round(P * S / t_all / 1e6, 4), "cores": cores, "sample": f"whole frame, OpenMP, {t_all:.1f}s"}}
            out["parity"] = {"rows_checked": rows_checked, "mismatching_values": bad, "bar": "bit-exact u16 planes vs CPU oracle"}
            if bad:
                print(json.dumps(out))
                sys.exit(f"bench.py: GPU frame differs from the oracle in {bad} values -- timing is void")
        print(json.dumps(out), flush=True)

    rs.close()
    if world > 1:
        dist.destroy_process_group()
    if rehearsal_bad:  # same rule as N=1: a frame that differs from the oracle voids the run (after the group is gone, so no rank hangs)
        sys.exit(f"bench.py: rehearsal frame differs from the oracle in {rehearsal_bad} values")


if __name__ == "__main__":
    main()
