"""ffb6d_amd -- MI355X-native (gfx950) hot path of FFB6D: exact batched KNN, the RandLA-Net
neighbour ops and the pixel<->point fusion gathers as hand-written HIP kernels behind a
C ABI (include/ffb6d_knn.h, include/ffb6d_ops.h), plus the host-side mirror of the
reference's Python operator interface.

Importing this package does not touch the GPU; the shared library is loaded on first
use and its absence is an error (no CPU fallback)."""

__version__ = "0.1.0"
