# the whole default bench (all variants) and the stereo line under every resident form of asd_track_frame: nothing hangs, nothing errors
set -e
mkdir -p gpurun_out/r4a
for e in 2 1 3; do
  ASD_CHAIN_EARLY=$e timeout -k 10 300 python3 bench.py --cpu-frames 0 > gpurun_out/r4a/full_e$e.json 2> gpurun_out/r4a/full_e$e.err || { tail -5 gpurun_out/r4a/full_e$e.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/r4a/full_e$e.json').read().strip().splitlines()[-1]); print('early=$e', round(d['value'],1), {k: round(d[k]['value'],1) for k in ('lane_variant','h2d_variant','do_mapping_variant') if d.get(k)})"
done
ASD_CHAIN_EARLY=2 timeout -k 10 300 python3 bench.py --workload euroc-stereo --steps 100 --warmup 30 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stereo early=2', round(d['value'],1))"
