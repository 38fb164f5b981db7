#!/bin/bash
# GPU session r03r: K1 with source batches in two named SGPR sets (loop unrolled by two, no loop-end copies) vs the product.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03r
mkdir -p $O
AB=$O/k1_sgpr_ab.txt
echo "# bench.py --steps 8 --warmup 2 (N = 2^20 fp32), one device, alternating: in-tree (rolled loop, 16 s_mov_b64 per batch) vs -DNB_K1_SGPR_AB=1" > $AB
run() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-spot --no-live-pmc $2 $3 $4 $5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))" >> $AB; tail -1 $AB; }
for i in 1 2 3; do
  run "rolled(in-tree)"
  run "ab_sets" --lib bench/ab/libnbody_sgpr_ab.so
done
run "acc64 rolled(in-tree)" --precision f32acc64
run "acc64 ab_sets" --precision f32acc64 --lib bench/ab/libnbody_sgpr_ab.so
run "js16 rolled(in-tree)" --j-split 16
run "js16 ab_sets" --j-split 16 --lib bench/ab/libnbody_sgpr_ab.so

