"""opencl_render_amd -- MI355X (gfx950) ray-trace + shade core behind the reference's RaytraceAll C ABI.

Contents: ``csrc/`` (HIP kernels + the C-ABI shared library ``libraytrace_hip.so``), ``raytrace`` (ctypes binding
mirroring the reference interface, ``source/opencl/raytrace.h``), ``scene`` (ABI-layout scene container and seeded
synthetic soups), ``tiles`` (multi-GPU tile partition + RCCL gather).
"""
from . import scene  # noqa: F401
from . import raytrace  # noqa: F401

__all__ = ["scene", "raytrace"]
