"""blitzar_amd: MI355X-native MSM / Pedersen-commitment engine behind Blitzar's C ABI.

The product is `blitzar_amd/lib/libblitzar_amd.so` (HIP kernels for gfx950 + the `sxt_*` C ABI,
sources under blitzar_amd/csrc).  `blitzar_amd.api` is a thin ctypes mirror of that ABI used by the
tests and bench.py.
"""
__all__ = ["api", "build"]
