"""GPU end-to-end case for the front-end and the sinks (run with -m gpu): mesh scene -> both lists built ON THE DEVICE ->
frame through the drop-in RaytraceAll on the GPU -> BMP/PPM bytes, compared with the bytes the same sinks produce from the CPU
oracle's planes of the same scene (bit-exact planes => identical files)."""
import os

import numpy as np
import pytest

from opencl_render_amd import demo as FC
import oracle_lib as O
from opencl_render_amd import frontend as F, raytrace as R

pytestmark = pytest.mark.gpu


def test_mesh_to_image_file_on_the_gpu(tmp_path, hip_lib):
    if hip_lib.rtHipDeviceCount() < 1:
        pytest.fail("no HIP device (the product has no CPU fallback)")
    sc = FC.room_scene(200, 150, samples=3)
    R.build_camera_list_device(sc, 0)
    R.build_scene_grid_device(sc, 0)
    ok, r, g, b = R.raytrace_all(1, sc)
    assert ok, R.last_error()
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    for ch, got, exp in zip("RGB", (r, g, b), want):
        assert np.array_equal(got, exp), f"plane {ch}: {(got != exp).sum()} values differ from the oracle"
    assert (r > 0).mean() > 0.95 and len(np.unique(r >> 8)) > 50  # a picture, not a blank or saturated frame
    for name, writer in (("gpu.bmp", F.write_bmp), ("gpu.ppm", F.write_ppm)):
        got_path, want_path = str(tmp_path / name), str(tmp_path / ("oracle_" + name))
        writer(got_path, r, g, b)
        writer(want_path, *want)
        data = open(got_path, "rb").read()
        assert data == open(want_path, "rb").read() and len(data) > 200 * 150 * 3


def test_command_line_harness_writes_the_image(tmp_path, hip_lib):
    """python -m opencl_render_amd: the dialog's fields on a command line (render.cpp:174-186) -> scene, lists on the device,
    RaytraceAll, image file."""
    from opencl_render_amd import __main__ as cli
    out = str(tmp_path / "room.bmp")
    assert cli.main(["--scene", "room", "--width", "160", "--height", "120", "--samples", "4", "--out", out]) == 0
    data = open(out, "rb").read()
    assert data[:2] == b"BM" and len(data) == 54 + 160 * 120 * 3
    sc = FC.room_scene(160, 120, samples=4)
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    want_path = str(tmp_path / "oracle.bmp")
    F.write_bmp(want_path, *want)
    assert data == open(want_path, "rb").read()
    R.lib().rtHipCacheClear()
