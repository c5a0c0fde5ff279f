"""rag_arc_amd — MI355X (gfx950) backend for RAG-ARC's dense-retrieval hot path.

Layout mirrors the reference's packages for the path:
    framework/                        AbstractConfig / AbstractModule / Register
    core/utils, core/retrieval, core/rerank
    encapsulation/embeddings, encapsulation/database/vector_db
    config/app_registration.py
    csrc/ + lib/librarc_hip.so        hand-written HIP kernels behind the C-ABI (include/rarc.h)
    hip/                              ctypes binding + HBM-resident flat index engine

Importing the package never touches the GPU; the first call into `rag_arc_amd.hip` that needs
the library loads it and fails loudly if it is missing.
"""
__version__ = "0.1.0"
