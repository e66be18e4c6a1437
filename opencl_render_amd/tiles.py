"""Image-tile partition across GPUs and the gather of finished tiles.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).  The frame is cut
into 128x128 tiles -- the reference's own NDRange granule (``source/opencl/raytrace.c:507-509``) -- dealt round-robin
to the ranks.  Every rank holds the full scene (secondary rays go anywhere) but only its tiles' slice of the per-pixel
candidate lists, renders into a compact ``[tile][plane][128*128]`` u16 buffer, and the buffers are gathered to rank 0,
which scatters them into the three row-major planes.  The RNG seed is a function of the GLOBAL pixel id, so the
partition is invisible in the result.

The only collective on the data path is that gather (one message per rank and frame, 96 KiB per tile); scene
distribution (``broadcast_scene``) happens once, outside the frame loop.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from .raytrace import TILE, tile_count, tiles_of_rank
from .scene import Scene

TILE_BYTES = 3 * TILE * TILE * 2


def max_tiles_per_rank(width: int, height: int, world: int) -> int:
    n = tile_count(width, height)
    return (n + world - 1) // world


def gather_tiles(local: torch.Tensor, width: int, height: int, rank: int, world: int, group=None,
                 force_collective: bool = False) -> Optional[torch.Tensor]:
    """Gathers the ranks' tile buffers (uint8 tensors of ntiles_r*TILE_BYTES bytes) to rank 0.

    Ranks may own one tile fewer than others; every message is padded to max_tiles_per_rank so the collective is a
    plain equal-size gather.  Returns on rank 0 a [world, max_tiles*TILE_BYTES] uint8 tensor, None elsewhere.
    A single rank has nothing to gather and gets its own buffer back; `force_collective` issues the collective all the
    same (tests run the RCCL path on a one-GPU box that way)."""
    cap = max_tiles_per_rank(width, height, world) * TILE_BYTES
    send = local
    if local.numel() != cap:
        send = torch.zeros(cap, dtype=torch.uint8, device=local.device)
        send[: local.numel()] = local
    if world == 1 and not force_collective:
        return send.view(1, cap)
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through host memory
        host = send.cpu()
        if rank == 0:
            out = torch.empty((world, cap), dtype=torch.uint8)
            dist.gather(host, list(out.unbind(0)), dst=0, group=group)
            return out.to(local.device)
        dist.gather(host, None, dst=0, group=group)
        return None
    if rank == 0:
        out = torch.empty((world, cap), dtype=torch.uint8, device=local.device)
        dist.gather(send, list(out.unbind(0)), dst=0, group=group)
        return out
    dist.gather(send, None, dst=0, group=group)
    return None


def detile_host(gathered: np.ndarray, width: int, height: int, world: int) -> List[np.ndarray]:
    """Host-side scatter of gathered tile buffers into [H,W] u16 planes (the device path uses rtHipDetile).
    `gathered` is [world, max_tiles*TILE_BYTES] uint8."""
    planes = [np.zeros((height, width), np.uint16) for _ in range(3)]
    tiles_x = (width + TILE - 1) // TILE
    for r in range(world):
        ids = tiles_of_rank(width, height, r, world)
        buf = np.ascontiguousarray(gathered[r]).view(np.uint16).reshape(-1, 3, TILE, TILE)
        for slot, t in enumerate(ids):
            ty, tx = divmod(int(t), tiles_x)
            y0, x0 = ty * TILE, tx * TILE
            h, w = min(TILE, height - y0), min(TILE, width - x0)
            for c in range(3):
                planes[c][y0:y0 + h, x0:x0 + w] = buf[slot, c, :h, :w]
    return planes


_ARRAY_FIELDS = ["eye", "eye_to_top_left", "left_to_right", "top_to_bottom", "vertex", "tri_index", "tri_material", "tri_uv",
                 "tri_normal", "mat_size", "mat_start", "textures", "light_type", "light_pos", "light_dir", "light_col",
                 "light_radius", "light_half_att", "cam_start", "cam_end", "cam_list", "box_min", "grid_start", "grid_list"]


_TORCH_DTYPES = {"<f4": torch.float32, "<i4": torch.int32, "<u4": torch.uint32, "|u1": torch.uint8}


def broadcast_scene(sc: Optional[Scene], rank: int, device: torch.device, group=None, rebuild_on_root: bool = False,
                    keep_on_device: bool = False) -> Scene:
    """Rank 0 holds the scene (and its lists); every other rank receives a copy.  Arrays travel as byte tensors on
    `device` -- over xGMI with the nccl backend -- instead of every rank rebuilding or re-reading them
    ('upload once via root then broadcast').  `rebuild_on_root`: rank 0 too returns a scene made from the
    broadcast tensors instead of the one it was given (tests: the bytes that travelled are the bytes that are used).
    `keep_on_device` (nccl only): a receiving rank's scene holds the broadcast tensors themselves, viewed in their dtypes --
    ResidentScene then hands the library device pointers (rtHipSceneDesc::arraysOnDevice) and nothing bounces through host memory."""
    head = [None]
    if rank == 0:
        scalars = dict(width=sc.width, height=sc.height, pixel_size_inv=sc.pixel_size_inv, sample_count=sc.sample_count,
                       name=sc.name, meta=sc.meta)
        head = [(scalars, {k: (getattr(sc, k).shape, getattr(sc, k).dtype.str) for k in _ARRAY_FIELDS})]
    dist.broadcast_object_list(head, src=0, group=group)
    scalars, layout = head[0]
    arrays = {}
    for k in _ARRAY_FIELDS:
        shape, dtype = layout[k]
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if dist.get_backend(group) == "gloo":
            device = torch.device("cpu")  # rehearsal mode: plain host broadcast
        if rank == 0:
            src = np.ascontiguousarray(getattr(sc, k))
            t = torch.from_numpy(src.view(np.uint8).reshape(-1).copy()).to(device)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if nbytes:
            dist.broadcast(t, src=0, group=group)
        keep = rank == 0 and not rebuild_on_root
        if keep:
            arrays[k] = getattr(sc, k)
        elif keep_on_device and t.is_cuda:
            arrays[k] = t.view(_TORCH_DTYPES[np.dtype(dtype).str]).reshape(shape)
        else:
            arrays[k] = t.cpu().numpy().view(np.dtype(dtype)).reshape(shape).copy()
        del t
    if rank == 0 and not rebuild_on_root:
        return sc
    return Scene(**scalars, **arrays)


class _DevicePointer:
    """Wraps a raw device pointer so torch can alias it (``torch.as_tensor`` via ``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def alias_device_bytes(ptr: int, nbytes: int, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_DevicePointer(ptr, nbytes), device=device)
