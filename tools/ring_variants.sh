#!/bin/bash
# tuning aid: libasdhip variants that differ in asdnet_ring.hip only (ASD_RING_ABL bits / other -D flags), linked against the objects of the
# default build.  usage: tools/ring_variants.sh name "-DASD_RING_ABL=4" [name2 "flags2" ...]  ->  asd-slam_amd/libasdhip_<name>.so
set -e
cd "$(dirname "$0")/../asd-slam_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  mkdir -p build_rv
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $flags -c asdnet_ring.hip -o build_rv/asdnet_ring_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libasdhip_$name.so build_rv/asdnet_ring_$name.o build/capi.o build/asdnet.o build/frontend.o build/quadtree.o build/matcher.o build/ba.o build/mapping.o build/bow.o
done
