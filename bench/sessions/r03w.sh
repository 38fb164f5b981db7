#!/bin/bash
# GPU session r03w: K1-f64 (nbody_force_f64_large, plain NB_F64 steps of large systems) — targets per lane R and sources per
# scalar-load batch U; bin/nbody_bench N 4 1 f64 against builds in bench/ab (LD_LIBRARY_PATH precedes the RUNPATH).
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03w
mkdir -p $O
T=$O/k1f64_blocking.txt
echo "# bin/nbody_bench N 4 1 f64: pairs/s; product = R 2, U 4" > $T
for n in 65536 262144; do
  for rep in 1 2; do
    for v in product r3 r4 r2u8 r4u2; do
      if [ $v = product ]; then L=""; else L="$PWD/bench/ab/$v"; fi
      r=$(LD_LIBRARY_PATH=$L ./bin/nbody_bench $n 4 1 f64 2>/dev/null | tail -1)
      echo "n=$n $v $r" | tee -a $T
    done
  done
done
