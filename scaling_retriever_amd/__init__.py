"""MI355X-native encode + retrieval hot path of scaling-retriever.

Host-side mirror of the reference's Python interface (LlamaBiDense / LlamaBiSparse,
DenseFlatIndexer, SparseRetrieval, store_embs ...) over libsr_hip.so, a C-ABI library
of hand-written HIP kernels for gfx950.  See DESIGN.md and include/sr_hip.h.
"""
__version__ = "0.1.0"
