"""MI355X-native CABAC bin codec (hot path of p-sawicki/entropy_coding: arithmetic bin
encoder/decoder + context model + binarisation) behind a C ABI (include/cabac_hip.h).

Python here is plumbing for tests and bench.py only: a ctypes binding (`capi`), the hipcc build
driver (`build`), the synthetic workloads of SURVEY.md §8(d) (`workload`) and the substream
sharding helpers for one-process-per-GPU runs (`sharding`).  The product is libcabac_hip.so and
the C++ host shim under host/."""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
