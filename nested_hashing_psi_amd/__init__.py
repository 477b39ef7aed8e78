"""nested_hashing_psi_amd -- MI355X (gfx950) implementation of the server-side batched-FHE
private-indexed-equality evaluation of SAP/nested-hashing-psi.

The product is libpiehip.so (C ABI in include/piehip.h, hand-written HIP kernels in csrc/).
This package is the thin Python host mirror used by the tests and bench.py: it loads the
library with ctypes and exposes the reference operator's interface
(src/Common/Crypto/PrivateIndexedEqualityCheck/BatchedFHEHIPPIE.hpp:18-49).
There is no CPU fallback: importing .pie without the built library raises.
"""
from .build import build  # noqa: F401

__all__ = ["build"]
