# One-box A/B of the ASDNet forward between two builds of the library: tools/asdnet_ab.sh LIB_A LIB_B  (paths; alternating, three rounds)
for r in 1 2 3; do
  for lib in "$@"; do
    echo "$lib: $(ASDHIP_LIB=$lib python3 tools/time_asdnet.py 2000 20 2>&1 | tail -2 | tr '\n' ' ')"
  done
done
