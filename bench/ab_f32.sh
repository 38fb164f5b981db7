#!/bin/bash
# bench/ab_f32.sh "<wg> <targets_per_lane> <j_split>" ... — interleaved A/B of launch shapes in ONE job on ONE device
# (devices differ by several percent, so numbers from different gpurun boxes must not be compared).
# env: BODIES (default 2^20), ROUNDS (2), STEPS (6)
cd "$(dirname "$0")/.." || exit 1
BODIES=${BODIES:-1048576}; ROUNDS=${ROUNDS:-2}; STEPS=${STEPS:-6}
for r in $(seq $ROUNDS); do
  for cfg in "$@"; do
    read -r wg tpl js <<< "$cfg"
    python bench.py --bodies $BODIES --steps $STEPS --warmup 1 --no-cpu-baseline --no-parity-spot --no-live-pmc --wg-size $wg --targets-per-lane $tpl --j-split $js 2>/dev/null \
      | python -c "import json,sys; d=json.load(sys.stdin); print('round $r  wg=$wg R=$tpl js=$js  %.4e pairs/s  %.2f%%  %.2f ms' % (d['value'], 100*d['roofline']['frac'], d['ms_per_step']))"
  done
done
