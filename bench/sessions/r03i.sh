#!/bin/bash
# GPU session r03i: per-node |sin| at a capture-time address (one dependent load less in the step kernel) — parity tests of
# everything that goes through the per-step engine, then nb_solve timelines and whole-program walls against round 2's
# binaries (bench/ab/r02 == this tree's product build before the change, profiles/r03_step_stamps_cost.txt) on one box.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03i
mkdir -p $O
python3 -m pytest tests/test_gpu_f64_parity.py tests/test_gpu_solve_schedule.py tests/test_gpu_scenarios_edge.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
hw5of() { case $1 in r02) echo bench/ab/r02/bin/hw5;; *) echo bin/hw5;; esac; }
for c in b200 b512 b1024; do
  for i in 1 2 3; do
    for v in r02 new; do
      NB_SOLVE_TRACE=1 $(hw5of $v) tests/golden/testcases/$c.in /tmp/t.$v.out 2>&1 | grep "scenarios done" | sed "s/^/$c $v /" | tee -a $O/solve_timeline.txt
      cmp -s /tmp/t.$v.out tests/golden/testcases/$c.out || echo "$c $v OUTPUT DIFFERS" | tee -a $O/solve_timeline.txt
    done
  done
done
python3 bench/replay_stamps.py b200 b1024 > $O/replay_stamps.txt 2>&1; grep -v Warning $O/replay_stamps.txt | tail -4
bash bench/run_testcases.sh > $O/testcases_wall.txt 2>&1; tail -14 $O/testcases_wall.txt
