#!/bin/bash
# bench/scaling.sh [bodies] [host: torch|native] — strong-scaling table of bench.py on the GPUs of ONE node (1, 2, 4, 8 ranks,
# RCCL), the run the driver performs at round end: `torch` = one process per GPU under torch.distributed.run (the driver's
# form), `native` = the command typed as is (one process, nb_sharded_*).  Needs that many GPUs; on a one-GPU box rehearse with
#   python bench.py --gpus 2 --exchange copy-one-gpu
#   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 bench.py --gpus 2 --backend gloo --single-device
cd "$(dirname "$0")/.." || exit 1
BODIES=${1:-1048576}
HOST=${2:-torch}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
base=""
for n in 1 2 4 8; do
  [ "$n" -gt "$NGPU" ] && break
  if [ "$n" -eq 1 ]; then out=$(python bench.py --gpus 1 --bodies $BODIES --no-cpu-baseline 2>/dev/null)
  elif [ "$HOST" = native ]; then out=$(python bench.py --gpus $n --bodies $BODIES 2>/dev/null | tail -1)
  else out=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
               bench.py --gpus $n --bodies $BODIES 2>/dev/null | tail -1); fi
  python - "$n" "$out" "$base" <<'PY'
import json, sys
n, d = int(sys.argv[1]), json.loads(sys.argv[2])
base = float(sys.argv[3]) if sys.argv[3] else d["value"]
print(f"{n} GPU(s): {d['value']:.4e} pairs/s  {d['ms_per_step']:.2f} ms/step  speedup {d['value'] / base:.2f}x  "
      f"per-rank kernel {100 * d['roofline']['frac']:.1f}% of peak")
PY
  [ -z "$base" ] && base=$(python -c "import json,sys; print(json.loads(sys.argv[1])['value'])" "$out")
done
