"""MI355X-native k-mer coverage normaliser: the per-read hot path of normalise_kmers_multi_large
(2-bit encode / canonical k-mer, k-mer table insert+count, per-read high-coverage counters, keep/skip test)
as hand-written HIP kernels behind a C ABI (include/nkml_hip.h), plus the thin host layers above it."""
from .binding import NkmlError, load_library, library_path  # noqa: F401
from .normaliser import KmerNormaliser  # noqa: F401

__all__ = ["KmerNormaliser", "NkmlError", "load_library", "library_path"]
