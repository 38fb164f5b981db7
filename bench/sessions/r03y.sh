#!/bin/bash
# GPU session r03y: context creation with one hipDeviceGetAttribute instead of hipGetDeviceProperties — nb_solve's
# "contexts created" stamp and whole-program wall, old (bench/ab/props) vs new, alternating.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03y
mkdir -p $O
for c in b20 b100 b200 b1024; do
  for i in 1 2 3; do
    for v in old new; do
      if [ $v = old ]; then H=bench/ab/props/bin/hw5; else H=bin/hw5; fi
      s=$(date +%s%N); NB_SOLVE_TRACE=1 $H tests/golden/testcases/$c.in /tmp/t.out 2> /tmp/trace.txt; e=$(date +%s%N)
      echo "$c $v wall $(( (e-s)/1000000 )) ms; $(grep 'contexts created' /tmp/trace.txt | head -1 | sed 's/\[nb_solve\] *//')" | tee -a $O/ctx_create.txt
    done
  done
done
