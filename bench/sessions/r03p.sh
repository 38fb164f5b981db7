#!/bin/bash
# GPU session r03p: workgroup size of the per-step fp64 kernel (S = 64): 128 / 256 (product) / 512 / 1024 threads =
# 2 / 4 / 8 / 16 targets per workgroup sharing one LDS stage of the system.  Single chain us/step and nb_solve timelines.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03p
mkdir -p $O
for w in 128 512 1024; do
  echo "== in-tree (256) vs $w" | tee -a $O/k2_wg_ab.txt
  python3 bench/k2_ab.py bench/ab/k2wg$w/nthu_ipc_nbody-simulation_amd/libnbody_amd.so 2>&1 | grep -v amdgpu | tee -a $O/k2_wg_ab.txt
done
for c in b200 b512 b1024; do
  for v in 256 128 512 1024 256; do
    if [ $v = 256 ]; then H=bin/hw5; else H=bench/ab/k2wg$v/bin/hw5; fi
    NB_SOLVE_TRACE=1 $H tests/golden/testcases/$c.in /tmp/t.$v.out 2>&1 | grep "scenarios done" | sed "s/^/$c wg$v /" | tee -a $O/solve_timeline.txt
    cmp -s /tmp/t.$v.out tests/golden/testcases/$c.out || echo "$c wg$v OUTPUT DIFFERS" | tee -a $O/solve_timeline.txt
  done
done
