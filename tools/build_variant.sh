#!/bin/bash
# tools/build_variant.sh <tag> <extra hipcc flags...>: the library built with extra -D switches, as satdump_amd/lib/libsdhip_<tag>.so (A/B on the GPU with
# SDHIP_LIB=...; the variants travel with the tree like the product library, and are git-ignored like it)
tag=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-result "$@" satdump_amd/csrc/*.hip -o satdump_amd/lib/libsdhip_$tag.so
