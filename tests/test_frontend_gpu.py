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


def test_obj_file_to_image_file_on_the_gpu(tmp_path, hip_lib):
    """The tool's path for a host without Cinema 4D: tests/data/scene.obj (quads, triangles, a fanned pentagon, an MTL with a PPM
    colour map, a BMP bump map, a transparent + reflective material and an emissive one) -> front-end arrays -> lists on the device ->
    RaytraceAll -> BMP, through the command line; the planes must be the oracle's for the same arrays, the file the one the sinks
    write from the oracle's planes."""
    from conftest import ROOT
    from opencl_render_amd import __main__ as cli, scene as S
    obj = os.path.join(ROOT, "tests", "data", "scene.obj")
    out = str(tmp_path / "obj.bmp")
    argv = ["--obj", obj, "--eye", "2.6", "2.2", "-3.4", "--look-at", "0", "0.4", "0", "--fov", "55", "--width", "192", "--height", "128",
            "--samples", "3", "--out", out]
    assert cli.main(argv) == 0
    data = open(out, "rb").read()
    mesh, materials = F.read_obj(obj)
    sc = F.scene_from_meshes([mesh], materials, [dict(type=S.LIGHT_DISTANT, dir=(0.3, -0.8, 0.5))], (2.6, 2.2, -3.4), (0, 0.4, 0), (0, 1, 0),
                             np.radians(55.0), 192, 128, samples=3)
    assert sc.triangle_count == 6 * 2 + 4 + 3 and sc.material_count == 4
    R.build_lists(sc)
    want = O.oracle_render(sc, threads=os.cpu_count() or 1)
    want_path = str(tmp_path / "oracle.bmp")
    F.write_bmp(want_path, *want)
    assert data == open(want_path, "rb").read()
    rgb = F.read_image(out)                       # the library reads back its own file
    assert np.array_equal(rgb, F.planes_to_rgb8(*want))
    assert (rgb.reshape(-1, 3).max(axis=1) > 0).mean() > 0.5 and len(np.unique(rgb[:, :, 0])) > 40  # a picture: floor texture, box, pyramid, sign
    R.lib().rtHipCacheClear()
