"""mmseqs2_amd - MI355X-native prefilter -> align hot path of MMseqs2.

The product is the C-ABI shared library ``mmseqs2_amd/lib/libmmgpu.so`` (hand-written HIP for gfx950,
sources in ``mmseqs2_amd/csrc``, contract in ``include/mmgpu.h``).  This Python package is plumbing for the
tests and for bench.py: a ctypes binding (``capi``) and the seeded workload generators (``workloads``).
There is no CPU fallback anywhere in this package: if the library is missing or no GPU is visible the
calls raise.
"""
from .capi import MMGpu, MMGpuError, build_library, library_path  # noqa: F401
