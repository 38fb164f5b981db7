#!/bin/bash
# bench/run_testcases.sh — the reference's whole checker loop (hw5.cu:618-629) on this build: every testcases/bN.in through
# bin/hw5, output compared byte for byte with the golden bN.out, wall time per case (process start + HIP init included).
cd "$(dirname "$0")/.." || exit 1
T=tests/golden/testcases
tmp=$(mktemp -d)
total_start=$(date +%s.%N)
for c in b20 b30 b40 b50 b60 b70 b80 b90 b100 b200 b512 b1024; do
  s=$(date +%s.%N)
  ./bin/hw5 $T/$c.in $tmp/$c.out || { echo "$c FAILED to run"; continue; }
  e=$(date +%s.%N)
  if cmp -s $tmp/$c.out $T/$c.out; then r=byte-identical; else r=DIFFERENT; fi
  python3 -c "print('%-6s %s  %.2f s' % ('$c', '$r', $e - $s))"
done
python3 -c "import time; print('total %.2f s' % ($(date +%s.%N) - $total_start))"
rm -rf $tmp
