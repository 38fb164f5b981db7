#!/bin/bash
# GPU session r03z: 1000 sustained steps (configs[4]'s step count and arithmetic, fp32 pair math / fp64 accumulate) at N = 2^20 on
# one GPU with checkpoints every 250 steps and conservation numbers; and 100 steps at N = 2^22 (configs[3]'s size), fp32.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03z
mkdir -p $O
python3 bench.py --bodies 1048576 --precision f32acc64 --steps 1000 --warmup 0 --report-every 100 --checkpoint /tmp/ck20.nbst \
    --checkpoint-every 250 --no-cpu-baseline --no-live-pmc --conservation > $O/n2e20_1000steps.json 2> $O/n2e20_1000steps.err
tail -3 $O/n2e20_1000steps.err
python3 bench.py --bodies 4194304 --precision f32 --steps 100 --warmup 0 --report-every 10 --no-cpu-baseline --no-live-pmc \
    --conservation > $O/n2e22_100steps.json 2> $O/n2e22_100steps.err
tail -3 $O/n2e22_100steps.err
python3 - <<'PY'
import json
for f in ("n2e20_1000steps", "n2e22_100steps"):
    d = json.loads(open(f"gpurun_out/r03z/{f}.json").read().strip().splitlines()[-1])
    c = d["conservation"]
    print(f, d["value"], d["steps"], d["ms_per_step"], d["roofline"]["frac"], c["momentum_drift_over_scale"], c["energy_rel_change_sampled"],
          d.get("checkpoints"), d["parity_spot"]["max_err_over_sum_abs"])
PY
