# quick A/B on one box (gpurun_out/r4a): headline runs under a few switches
O=gpurun_out/r4a; mkdir -p $O
run() { timeout -k 10 200 python3 bench.py --cpu-frames 0 --no-lane-variant --no-h2d-variant --no-do-mapping-variant --steps 450 --warmup 60 "$@" 2>> $O/bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['steady_state']; print(round(d['value'],1), round(s['ms_tracking_per_frame'],3), round(s['ms_per_local_ba'],3), round(s['ms_waiting_for_extractor_per_frame'],3), round(d['roofline']['asdnet_forward_ms'],3))"; }
echo "solve mid prep mid   $(run)"
echo "solve hi  prep mid   $(ASD_SOLVE_PRIO=2 run)"
echo "solve mid prep hi    $(ASD_PREP_PRIO=2 run)"
echo "solve hi  prep hi    $(ASD_SOLVE_PRIO=2 ASD_PREP_PRIO=2 run)"
echo "solve lo  prep mid   $(ASD_SOLVE_PRIO=0 run)"
echo "inorder   prep mid   $(ASD_CHAIN_EARLY=0 run)"
echo "inorder   prep hi    $(ASD_CHAIN_EARLY=0 ASD_PREP_PRIO=2 run)"
