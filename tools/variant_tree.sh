#!/bin/bash
# tools/variant_tree.sh NAME SRC "FLAGS": a tree _NAME (bench.py + package + libraries) whose libasdhip.so has SRC (matcher.hip, ba.hip, frontend.hip, ...)
# compiled with extra FLAGS, every other object taken from the current build -- for one-box A/Bs with tools/ab.sh "x:ASD_REPO=_NAME".
set -e
NAME=$1; SRC=$2; FLAGS=$3
cd "$(dirname "$0")/../asd-slam_amd/csrc"
base=${SRC%.*}
contract="-ffp-contract=off"; case $SRC in asdnet.hip|asdnet_ring.hip|ba.hip) contract="";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize $contract $FLAGS -c $SRC -o build/${base}_${NAME}_tmp.o
T=../../_$NAME; rm -rf $T; mkdir -p $T/asd-slam_amd
objs=""; for o in capi asdnet asdnet_ring frontend quadtree matcher ba mapping bow; do if [ $o = $base ]; then objs="$objs build/${base}_${NAME}_tmp.o"; else objs="$objs build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/asd-slam_amd/libasdhip.so $objs
cp ../*.py ../libasdtrack.so ../libasdhip_s32.so $T/asd-slam_amd/; cp ../../bench.py ../../__graft_entry__.py $T/
rm -f build/${base}_${NAME}_tmp.o

echo "built _$NAME"
