#!/bin/bash
# GPU session r03e: K1 A/B of four builds on one device (pair grouping x waves-per-SIMD bound), alternating; then the
# smallest-graph bracket of the rocprofv3 hipGraphLaunch crash and, if any size survives, one trace of hw5's replay path.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
AB=$O/k1_ab.txt
echo "# bench.py --steps 8 --warmup 2 (N = 2^20 fp32), one device, alternating builds; columns: build, ms/step, kernel ms, frac of 157.3 TF" > $AB
run() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-spot $2 $3 $4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))" >> $AB; tail -1 $AB; }
for i in 1 2 3; do
  run "group2_waves4(in-tree)"
  run "group1_waves2(rounds1-2)" --lib bench/ab/libnbody_pairorder1.so
  run "group2_waves2" --lib bench/ab/libnbody_group2_waves2.so
  run "group1_waves4" --lib bench/ab/libnbody_group1_waves4.so
done
run "acc64 group2_waves4(in-tree)" --precision f32acc64
run "acc64 group1_waves2(rounds1-2)" --precision f32acc64 --lib bench/ab/libnbody_pairorder1.so
python3 -m pytest tests/test_gpu_f32_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
L=$O/graph_trace_limit.txt
: > $L
OKG=0
for G in 2 4 8 16 32 64 100; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/lr_$G -o lr -- ./bench/ubench/launch_rate $G quick > $O/lr_$G.log 2>&1
  rc=$?
  echo "launch_rate, graph of $G kernel nodes, under rocprofv3 --kernel-trace --stats: rc=$rc; $(grep -c SIGSEGV $O/lr_$G.log) SIGSEGV; $(grep 'graph(' $O/lr_$G.log)" | tee -a $L
  [ $rc -ne 0 ] && break
  OKG=$G
done
echo "largest graph that survived the tracer: $OKG nodes" | tee -a $L
if [ $OKG -ge 2 ]; then
  NB_GRAPH_CHUNK=$OKG NB_HW5_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/replay_b1024 -o replay -- ./bin/hw5 tests/golden/testcases/b1024.in /tmp/b1024.out > $O/replay_b1024.log 2>&1
  echo "hw5 b1024 with NB_GRAPH_CHUNK=$OKG under the tracer: rc=$?" | tee -a $L
  cmp /tmp/b1024.out tests/golden/testcases/b1024.out && echo "b1024 output identical under the tracer" | tee -a $L
  find $O/replay_b1024 -name "*kernel_stats.csv" -exec cp {} $O/replay_b1024_kernel_stats.csv \;
fi
find $O -name "*.csv" -size +1M -delete
