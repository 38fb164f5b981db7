#!/bin/bash
# GPU session r03c: K1 pair-order A/B (same device, alternating), rocprofv3 evidence of the new kernel, one trace of the
# graph-replay path, the node-count bracket of the rocprofv3 hipGraphLaunch crash, kernel parity tests.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
AB=$O/pair_order_ab.txt
echo "# bench.py --steps 8 --warmup 2, same device, alternating: in-tree lib (pairs in groups of 2) vs bench/ab/libnbody_pairorder1.so (pair after pair, rounds 1-2)" > $AB
for i in 1 2 3; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-spot 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('group2 ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $AB
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity-spot --lib bench/ab/libnbody_pairorder1.so 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('group1 ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $AB
done
cat $AB
# acc64 too
for lib in "" "--lib bench/ab/libnbody_pairorder1.so"; do
  python3 bench.py --steps 4 --warmup 1 --precision f32acc64 --no-cpu-baseline --no-parity-spot $lib 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('acc64 $lib', d['ms_per_step'], d['roofline']['frac'])" >> $AB
done
tail -2 $AB
python3 -m pytest tests/test_gpu_f32_parity.py tests/test_gpu_sharded_native.py tests/test_gpu_distributed.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
python3 bench.py --steps 10 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
bash bench/profile.sh r03 > $O/profile.log 2>&1; tail -25 $O/profile.log
# the graph-replay path under the tracer: 100-node graphs, normal process exit so the tool can flush
NB_GRAPH_CHUNK=100 NB_HW5_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/replay_b1024 -o replay -- ./bin/hw5 tests/golden/testcases/b1024.in /tmp/b1024.out > $O/replay_b1024.log 2>&1
echo "replay trace rc=$?" | tee -a $O/replay_b1024.log
cmp /tmp/b1024.out tests/golden/testcases/b1024.out && echo "b1024 output identical under the tracer" | tee -a $O/replay_b1024.log
# node-count bracket of the tracer's crash inside hipGraphLaunch (ascending; stops at the first failure)
L=$O/graph_trace_limit.txt
: > $L
for G in 150 250 400 600 800; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/lr_$G -o lr -- ./bench/ubench/launch_rate $G quick > $O/lr_$G.log 2>&1
  rc=$?
  echo "launch_rate graph of $G nodes under rocprofv3 --kernel-trace --stats: rc=$rc $(grep -c SIGSEGV $O/lr_$G.log) SIGSEGV lines; $(grep 'graph(' $O/lr_$G.log)" | tee -a $L
  [ $rc -ne 0 ] && break
done
find $O -name "*.csv" -size +2M -delete
