#!/bin/bash
# A/B builds of the ASDNet kernels: tools/build_variant.sh <name> <defs...> -> asd-slam_amd/libasdhip_<name>.so (only asdnet.hip is
# recompiled; select it with ASDHIP_LIB=asd-slam_amd/libasdhip_<name>.so).  The variant libraries are git-ignored.
set -e
cd "$(dirname "$0")/../asd-slam_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize "$@" -c asdnet.hip -o build/asdnet_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libasdhip_$name.so build/asdnet_$name.o build/capi.o build/frontend.o build/quadtree.o build/matcher.o build/ba.o build/mapping.o build/bow.o
