#!/bin/bash
# GPU session r03x: K3 (persistent small-n engine) with 16 instead of 8 lanes per target for n <= 64: nb_solve timelines and
# golden outputs, product vs bench/ab/k3s16 (-DNB_K3_S16_MAX=64), alternating.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03x
mkdir -p $O
for c in b20 b30 b40 b50 b60; do
  for i in 1 2 3; do
    for v in s8 s16; do
      if [ $v = s8 ]; then H=bin/hw5; else H=bench/ab/k3s16/bin/hw5; fi
      NB_SOLVE_TRACE=1 $H tests/golden/testcases/$c.in /tmp/t.$v.out 2>&1 | grep "first wave" | sed "s/^/$c $v /" | tee -a $O/k3_s16.txt
      cmp -s /tmp/t.$v.out tests/golden/testcases/$c.out || echo "$c $v OUTPUT DIFFERS" | tee -a $O/k3_s16.txt
    done
  done
done
