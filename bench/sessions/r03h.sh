#!/bin/bash
# GPU session r03h: whole-checker wall time and nb_solve timeline of this tree (product build), of the instrumented
# stamps build and of round 2's binaries (bench/ab/r02, commit 9d27bb6) on the same box, alternating; replay-path stamps.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03h
mkdir -p $O
W=$O/hw5_r02_vs_r03.txt
: > $W
hw5of() { case $1 in r02) echo bench/ab/r02/bin/hw5;; stamps) echo bench/ab/stamps/bin/hw5;; *) echo bin/hw5;; esac; }
for c in b100 b200 b512 b1024; do
  for i in 1 2 3; do
    for v in r02 r03 stamps; do
      H=$(hw5of $v)
      s=$(date +%s%N); $H tests/golden/testcases/$c.in /tmp/$c.$v.out; e=$(date +%s%N)
      cmp -s /tmp/$c.$v.out tests/golden/testcases/$c.out && ok=identical || ok=DIFFERENT
      echo "$c $v $(( (e - s) / 1000000 )) ms $ok" >> $W
    done
  done
done
cat $W
for c in b200 b512 b1024; do
  for v in r02 r03 stamps; do
    NB_SOLVE_TRACE=1 $(hw5of $v) tests/golden/testcases/$c.in /tmp/t.out 2>&1 | grep "scenarios done" | sed "s/^/$c $v /" | tee -a $O/solve_timeline.txt
  done
done
python3 bench/replay_stamps.py b200 b512 b1024 > $O/replay_stamps.txt 2>&1; grep -v Warning $O/replay_stamps.txt | tail -8
python3 -m pytest tests/test_gpu_solve_schedule.py -m gpu -x -q -k "stamps or handoff or chunk" > $O/tests.log 2>&1; tail -3 $O/tests.log
