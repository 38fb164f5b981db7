#!/bin/bash
# GPU session r03g: device clock probe + step stamps, and the whole-checker wall time of this tree against round 2's
# binaries (bench/ab/r02, built from commit 9d27bb6) on the same box, alternating.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03g
mkdir -p $O
./bench/debug/clock_probe > $O/clock_probe.txt 2>&1; cat $O/clock_probe.txt
python3 bench/replay_stamps.py b200 b1024 > $O/replay_stamps.txt 2>&1; grep -v Warning $O/replay_stamps.txt | tail -12
W=$O/hw5_r02_vs_r03.txt
: > $W
for c in b100 b200 b512 b1024; do
  for i in 1 2 3; do
    for v in r02 r03; do
      if [ $v = r02 ]; then H=bench/ab/r02/bin/hw5; else H=bin/hw5; fi
      s=$(date +%s.%N); $H tests/golden/testcases/$c.in /tmp/$c.$v.out; e=$(date +%s.%N)
      cmp -s /tmp/$c.$v.out tests/golden/testcases/$c.out && ok=identical || ok=DIFFERENT
      echo "$c $v $(echo "$e - $s" | bc) s $ok" >> $W
    done
  done
done
cat $W
for v in r02 r03; do
  if [ $v = r02 ]; then H=bench/ab/r02/bin/hw5; else H=bin/hw5; fi
  NB_SOLVE_TRACE=1 $H tests/golden/testcases/b200.in /tmp/t.out 2> $O/trace_b200_$v.txt
  NB_SOLVE_TRACE=1 $H tests/golden/testcases/b512.in /tmp/t.out 2> $O/trace_b512_$v.txt
done
tail -6 $O/trace_b200_r02.txt $O/trace_b200_r03.txt
