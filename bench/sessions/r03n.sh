#!/bin/bash
# GPU session r03n: HBM traffic (live rocprofv3 --pmc passes inside bench.py) against speed for 1..32 source slices of K1,
# N = 2^20 fp32, one device — the measured trade-off behind "bytes are free here, issue slots are not".
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03n
mkdir -p $O
T=$O/traffic_vs_slices.txt
echo "# bench.py --steps 6 --warmup 2 --j-split J (512-thread workgroups, 8 targets per lane), live PMC: columns J, ms/step, frac of 157.3 TF, HBM MB/step (2*FETCH+WRITE), raw FETCH MB, WRITE MB, VALU busy" > $T
for J in 1 2 4 8 16 32; do
  python3 bench.py --steps 6 --warmup 2 --j-split $J --wg-size 512 --targets-per-lane 8 --no-cpu-baseline --no-parity-spot 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']; l = r.get('live_pmc') or {}
print($J, round(d['ms_per_step'], 2), round(r['frac'], 4), round((r['traffic'] or 0) / 1e6, 1), round(l.get('fetch_kib_raw', 0) * 1024 / 1e6, 1), round(l.get('write_kib', 0) * 1024 / 1e6, 1), round(r['valu_busy'] or 0, 3), r['kernel'])" >> $T
  tail -1 $T
done
