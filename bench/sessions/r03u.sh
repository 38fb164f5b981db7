#!/bin/bash
# GPU session r03u: K1 sliced launch with and without the per-tile s_barrier (256 sources) on the SGPR path.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03u
mkdir -p $O
AB=$O/k1_tile_barrier_ab.txt
echo "# bench.py --steps 8 --warmup 3 (N = 2^20 fp32), one device, alternating: in-tree (s_barrier per 256 sources) vs -DNB_K1_TILE_BARRIER=0" > $AB
run() { python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity-spot $2 $3 $4 $5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$1', round(d['ms_per_step'],3), round(r['frac'],4), 'traffic MB', round((r['traffic'] or 0)/1e6,1), 'valu_busy', round(r['valu_busy'] or 0,3))" >> $AB; tail -1 $AB; }
run "warmup(in-tree)" --no-live-pmc
for i in 1 2 3; do
  run "barrier(in-tree)" --no-live-pmc
  run "no_barrier" --no-live-pmc --lib bench/ab/libnbody_nobarrier.so
done
run "barrier(in-tree) js1" --no-live-pmc --j-split 1 --wg-size 512 --targets-per-lane 8
run "no_barrier js1" --no-live-pmc --j-split 1 --wg-size 512 --targets-per-lane 8 --lib bench/ab/libnbody_nobarrier.so
