#!/bin/bash
# Developer build of libmmrag.so with extra flags for ONE translation unit, next to the product library:
#   tools/build_variant.sh <tag> <source.hip> <flags...>   ->  multimodal_rag_amd/lib/libmmrag_<tag>.so
# (the other objects are taken from the product build; tools/ab_search.py loads it through MMRAG_AB_LIB)
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
obj=multimodal_rag_amd/csrc/_obj
base=$(basename "$src" .hip)
hipcc "$@" -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I include -c "multimodal_rag_amd/csrc/$src" -o "$obj/${base}_$tag.o"
others=$(ls $obj/*.o | grep -v "/${base}\.o$" | grep -v "/${base}_[A-Za-z0-9]*\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o "multimodal_rag_amd/lib/libmmrag_$tag.so" $others "$obj/${base}_$tag.o"
echo "multimodal_rag_amd/lib/libmmrag_$tag.so"
