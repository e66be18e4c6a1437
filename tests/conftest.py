import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden_scene(name):
    """Rebuilds a Scene (inputs) and the expected planes from tests/golden/scene_<name>.npz."""
    from opencl_render_amd.scene import Scene
    z = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
    w, h, s = (int(v) for v in z["dims"])
    counts = np.zeros(256 ** 3 + 1, np.uint32)
    counts[z["grid_cells"].astype(np.int64) + 1] = z["grid_counts"]
    grid_start = np.cumsum(counts, dtype=np.uint64).astype(np.uint32)
    sc = Scene(width=w, height=h, eye=z["eye"], eye_to_top_left=z["eye_to_top_left"], left_to_right=z["left_to_right"],
               top_to_bottom=z["top_to_bottom"], pixel_size_inv=float(z["pixel_size_inv"]), sample_count=s,
               vertex=z["vertex"], tri_index=z["tri_index"], tri_material=z["tri_material"], tri_uv=z["tri_uv"],
               tri_normal=z["tri_normal"], mat_size=z["mat_size"], mat_start=z["mat_start"], textures=z["textures"],
               light_type=z["light_type"], light_pos=z["light_pos"], light_dir=z["light_dir"], light_col=z["light_col"],
               light_radius=z["light_radius"], light_half_att=z["light_half_att"], cam_start=z["cam_start"], cam_end=z["cam_end"],
               cam_list=z["cam_list"], box_min=z["box_min"], grid_start=grid_start, grid_list=z["grid_list"], name=name)
    for k, v in list(vars(sc).items()):
        if isinstance(v, np.ndarray):
            setattr(sc, k, np.ascontiguousarray(v))
    return sc, [z["out_r"], z["out_g"], z["out_b"]]


def golden_names():
    return sorted(f[len("scene_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("scene_") and f.endswith(".npz"))


@pytest.fixture(scope="session")
def hip_lib():
    from opencl_render_amd import raytrace
    return raytrace.lib()


@pytest.fixture(scope="session")
def gpu_count(hip_lib):
    return hip_lib.rtHipDeviceCount()
