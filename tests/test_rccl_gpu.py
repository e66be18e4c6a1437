"""The multi-GPU code path on its REAL backend, on the one GPU a test box has: `torch.distributed` with backend "nccl" (RCCL on
ROCm) at world size 1, inside the test process (no re-exec of a process that has touched the GPU).  Everything bench.py and
tiles.py do between ranks runs here for real -- scene broadcast (object list + byte tensors on the device), the tile gather on a
tensor that ALIASES the library's raw device pointer, a barrier, the float64 / int64 all-reduces -- with the frame rendered on the
stream the collectives are ordered behind, and the de-tiled planes compared with the golden (reference kernel) planes.
What a world of 1 cannot show is the wire: scaling stays unmeasured until a SCALE record exists."""
import os
import socket

import numpy as np
import pytest

from conftest import load_golden_scene
from opencl_render_amd import raytrace as R, tiles as T

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world1():
    import torch
    import torch.distributed as dist
    if R.lib().rtHipDeviceCount() < 1 or not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (the HIP path has no CPU fallback)")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        yield device
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["odd_size_multi_tile", "mirror_hall"])
def test_frame_through_broadcast_gather_and_detile_on_rccl(rccl_world1, name):
    import torch
    import torch.distributed as dist
    device = rccl_world1
    assert dist.get_backend() == "nccl"
    sc0, want = load_golden_scene(name)
    # object list + one device byte tensor per array, over RCCL; the scene keeps the tensors (as a receiving rank of bench.py does) and the
    # library copies them HBM to HBM (rtHipSceneDesc::arraysOnDevice)
    sc = T.broadcast_scene(sc0, 0, device, rebuild_on_root=True, keep_on_device=True)
    assert sc is not sc0 and sc.vertex.is_cuda and sc.grid_start.is_cuda and sc.vertex.shape == sc0.vertex.shape
    assert np.array_equal(sc.vertex.cpu().numpy(), sc0.vertex) and np.array_equal(sc.grid_list.cpu().numpy().view(np.uint32), sc0.grid_list)
    W, H, P = sc.width, sc.height, sc.pixels
    world, rank = 1, 0
    rs = R.ResidentScene(sc, 0, R.tiles_of_rank(W, H, rank, world))
    try:
        ptr, nbytes = rs.tile_buffer()
        tile_tensor = T.alias_device_bytes(ptr, nbytes, device)     # torch's view of the library's tile buffer (no copy)
        assert tile_tensor.data_ptr() == ptr
        work_stream = torch.cuda.Stream(device)
        planes = torch.zeros(3 * P * 2, dtype=torch.uint8, device=device)
        slots = T.max_tiles_per_rank(W, H, world)
        ids = torch.from_numpy(R.tiles_of_rank(W, H, 0, 1).astype(np.int32)).to(device)
        for frame in range(3):  # watched frame, then planned ones: the gather is enqueued right behind the frame's last kernel
            with torch.cuda.stream(work_stream):
                rs.render(work_stream.cuda_stream)
                gathered = T.gather_tiles(tile_tensor, W, H, rank, world, force_collective=True)
                assert gathered.shape == (1, slots * T.TILE_BYTES) and gathered.data_ptr() != ptr  # a real gather output
                base = planes.data_ptr()
                rc = R.lib().rtHipDetileStore(0, gathered.data_ptr(), ids.data_ptr(), slots, W, H, base, base + 2 * P, base + 4 * P,
                                              work_stream.cuda_stream)
                assert rc == 0, R.last_error()
            dist.barrier()
            torch.cuda.synchronize()
            assert not rs.finish()
            got = planes.cpu().numpy().view(np.uint16).reshape(3, H, W)
            for c in range(3):
                assert np.array_equal(got[c], want[c]), f"{name}: plane {c} of frame {frame} differs after the RCCL gather"
        # bench.py's reductions: the slowest rank's time (float64 MAX) and the summed work counters (int64 SUM), on the device
        t = torch.tensor([1.25], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        stats = rs.render_counted()
        keys = sorted(stats)
        v = torch.tensor([stats[k] for k in keys], dtype=torch.int64, device=device)
        dist.all_reduce(v)
        assert {k: int(x) for k, x in zip(keys, v.tolist())} == stats
    finally:
        rs.close()
