#!/bin/bash
# Same-box A/B of environment settings: bench.py alternately in fresh processes.
#   bash scripts/ab_env_bench.sh <reps> "<tag>:<VAR=val VAR2=val>" ...        (empty setting = defaults)
set -o pipefail
mkdir -p gpurun_out
REPS=$1; shift
FLAGS="--steps 20 --warmup 3 --no-side --no-cpu-baseline --no-grid --no-parity"
for rep in $(seq 1 $REPS); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    ( for kv in $envs; do export "$kv"; done; python3 bench.py $FLAGS 2>/dev/null > gpurun_out/ab_env_${tag}_$rep.json ) || echo "$tag failed"
  done
done
python3 - $REPS "$@" <<'PY'
import json, sys
reps = int(sys.argv[1])
for spec in sys.argv[2:]:
    tag = spec.split(":")[0]
    ms = []
    for rep in range(1, reps + 1):
        try:
            d = json.load(open("gpurun_out/ab_env_%s_%d.json" % (tag, rep)))
        except Exception:
            continue
        ms.append((d["ms_per_step"], d["conv_stack"]["unet_fwd_ms"], d["conv_stack"]["unet_bwd_ms"], d["roofline"]["avg_launch_us"]))
    if ms:
        print("%-12s ms/step %s | unet fwd %s | bwd %s | nn us %s" % ((tag,) + tuple(" ".join("%.2f" % v[i] for v in ms) for i in range(4))))
PY
