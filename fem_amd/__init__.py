"""fem_amd — MI355X-native implementation of FEM's per-read mapping hot path.

The product is native: `csrc/libfemhip.so` (HIP kernels behind the C ABI of include/fem_hip.h),
`csrc/libfemhost.so` (host-side C++: I/O, mapping tail, synthetic data) and the `csrc/FEM` command line.
This package is only the thin ctypes layer that tests and bench.py use to call them.
There is no CPU fallback: every mapping entry point raises if the HIP library or a GPU is missing.
"""
from .device import Device, FemError, Params, load_hip  # noqa: F401
