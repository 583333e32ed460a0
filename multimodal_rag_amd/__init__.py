"""MI355X-native embed-and-retrieve path for the multimodal_rag service.

Host side mirrors the reference's `app.utils` interface for the hot path
(`EmbeddingManager`, `MultiVectorRetriever`); all arithmetic runs in hand-written HIP
kernels (gfx950) behind the C-ABI of include/mmrag.h, loaded from
multimodal_rag_amd/lib/libmmrag.so.  There is no CPU fallback.
"""

__version__ = "0.1.0"
