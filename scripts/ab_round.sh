#!/bin/bash
# Same-box A/B of the whole step: another tree (default build_exp/prev = `git archive <commit>` exported and built there; round 4
# first used build_exp/r03 = the round-3 tree) against this tree, bench.py run alternately in fresh processes, three times each
# (boxes of the pool differ by +-1.5 %, so numbers from different gpurun calls do not rank builds).  Run on the GPU box from the
# repository root:  bash scripts/ab_round.sh [other-tree] [tag-of-other] [tag-of-this]
set -o pipefail
mkdir -p gpurun_out
OTHER=${1:-build_exp/prev}
TA=${2:-prev}
TB=${3:-this}
FLAGS="--steps 20 --warmup 3 --no-side --no-cpu-baseline --no-grid --no-parity"
for rep in 1 2 3; do
  (cd $OTHER && python3 bench.py $FLAGS 2>/dev/null) > gpurun_out/ab_round_${TA}_$rep.json || exit 1
  python3 bench.py $FLAGS 2>/dev/null > gpurun_out/ab_round_${TB}_$rep.json || exit 1
done
python3 - $TA $TB <<'PY'
import json, sys
for tag in sys.argv[1:3]:
    ms = []
    for rep in (1, 2, 3):
        d = json.load(open("gpurun_out/ab_round_%s_%d.json" % (tag, rep)))
        ms.append((d["ms_per_step"], d["conv_stack"]["unet_fwd_ms"], d["conv_stack"]["unet_bwd_ms"], d["roofline"]["avg_launch_us"]))
    print(tag, "ms/step %s | unet fwd %s | bwd %s | nn us %s" % tuple(" ".join("%.2f" % v[i] for v in ms) for i in range(4)))
PY
