#!/bin/bash
# One-box A/B of the headline bench under environment switches or bench flags (replaces round 4's twenty tools/ab_*.sh; the variants they
# measured -- resident solver forms, frustum tail, extractor hold, slices, polling, stream priorities -- were removed from the product in round 5,
# their results are in DESIGN.md section 4.0c).
#   tools/ab.sh [-r REPEAT] [-s STEPS] CASE [CASE ...]      CASE = "label:ENV1=a ENV2=b -- --bench-flag ..."   (env and flags both optional)
#   e.g.  gpurun -- 'bash tools/ab.sh "base:" "la3:ASD_BENCH_LOOKAHEAD=3" "chain: -- --chain" "old:ASD_REPO=_old"'
# ASD_REPO=<dir> runs bench.py of another checkout inside the repository (a tree built before a change), same box, same call.
# Prints per case: frames/s, steady-state tracking ms per frame, ms per LocalBA, ms waiting for the extractor per frame, ASDNet forward ms.
REPEAT=2; STEPS=450
while getopts "r:s:" o; do case $o in r) REPEAT=$OPTARG;; s) STEPS=$OPTARG;; esac; done; shift $((OPTIND - 1))
O=gpurun_out/ab; mkdir -p $O
for rep in $(seq $REPEAT); do
  for c in "$@"; do
    label=${c%%:*}; rest=${c#*:}; envs=${rest%%--*}; flags=""; [[ "$rest" == *"--"* ]] && flags=${rest#*--}
    repo=.; for kv in $envs; do [[ $kv == ASD_REPO=* ]] && repo=${kv#ASD_REPO=}; done
    extra=""; grep -q -- "--no-local-map-sweep" $repo/bench.py && extra="--no-local-map-sweep"
    out=$(cd $repo && env $envs timeout -k 10 300 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --no-one-submission-variant $extra \
          --steps $STEPS --warmup 60 $flags 2>> $OLDPWD/$O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']
print(round(d['value'],1), round(s['ms_tracking_per_frame'],3), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],3), round(d['roofline']['asdnet_forward_ms'],3))")
    printf "%-12s %s\n" "$label" "$out"
  done
done
