#!/bin/bash
# A/B builds of one kernel source: tools/build_variant.sh <name> <source stem: asdnet|matcher|ba|frontend> <defs...>
#   -> asd-slam_amd/libasdhip_<name>.so (only that source is recompiled; select it with ASDHIP_LIB=asd-slam_amd/libasdhip_<name>.so).
# The variant libraries are git-ignored.
set -e
cd "$(dirname "$0")/../asd-slam_amd/csrc"
name=$1; stem=$2; shift; shift
FP=""; case $stem in asdnet|ba) ;; *) FP="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $FP "$@" -c $stem.hip -o build/${stem}_$name.o
objs=""; for o in capi asdnet frontend quadtree matcher ba mapping bow; do if [ $o = $stem ]; then objs="$objs build/${stem}_$name.o"; else objs="$objs build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libasdhip_$name.so $objs
