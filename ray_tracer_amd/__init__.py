"""MI355X-native render hot path of mbStavola/ray_tracer.

`ray_tracer_amd.api` is the Python host over the C ABI (include/mirt.h);
`ray_tracer_amd.configs` holds the benchmark configurations.  The kernels live
in csrc/ (HIP, gfx950) and are reached only through csrc/libmirt.so.
"""
from . import _abi, api, configs  # noqa: F401

__all__ = ["api", "configs"]
