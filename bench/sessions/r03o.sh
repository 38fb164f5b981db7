#!/bin/bash
# GPU session r03o: barrier-free K2 body (S = 64, n <= 1024) vs the LDS-staged one (bench/ab/k2staged, -DNB_K2_DIRECT=0):
# parity tests of everything on the per-step engine, single-chain us/step, nb_solve timelines, whole-program walls.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03o
mkdir -p $O
python3 -m pytest tests/test_gpu_f64_parity.py tests/test_gpu_solve_schedule.py tests/test_gpu_scenarios_edge.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
python3 bench/k2_ab.py bench/ab/k2staged/nthu_ipc_nbody-simulation_amd/libnbody_amd.so > $O/k2_ab.txt 2>&1; grep -v amdgpu $O/k2_ab.txt
for c in b200 b512 b1024; do
  for i in 1 2 3; do
    for v in staged direct; do
      if [ $v = staged ]; then H=bench/ab/k2staged/bin/hw5; else H=bin/hw5; fi
      NB_SOLVE_TRACE=1 $H tests/golden/testcases/$c.in /tmp/t.$v.out 2>&1 | grep "scenarios done" | sed "s/^/$c $v /" | tee -a $O/solve_timeline.txt
      cmp -s /tmp/t.$v.out tests/golden/testcases/$c.out || echo "$c $v OUTPUT DIFFERS" | tee -a $O/solve_timeline.txt
    done
  done
done
python3 bench/replay_stamps.py b200 b1024 > $O/replay_stamps.txt 2>&1; grep "graph replay" $O/replay_stamps.txt
