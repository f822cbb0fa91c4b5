# tuning aid: per-layer times of the ring kernels under each variant library built by tools/ring_variants.sh
for v in "" $@; do
  lib=$PWD/asd-slam_amd/libasdhip${v:+_$v}.so
  echo "== ${v:-default}"
  ASDHIP_LIB=$lib ASD_ASDNET_RING=${RING:-7} timeout -k 10 120 python tools/time_asdnet.py 2000 20 2>&1 | tail -2 | head -1
done
