#!/bin/bash
# build libmmrag.so and fail loudly on any compiler error (use before every gpurun)
cd "$(dirname "$0")/.." && python -m multimodal_rag_amd.build > /tmp/mmrag_build.log 2>&1 || { grep -E "error" -A3 /tmp/mmrag_build.log | head -30; echo BUILD FAILED; exit 1; }
tail -1 /tmp/mmrag_build.log
