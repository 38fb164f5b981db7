#!/bin/bash
# GPU session r03f: full GPU test suite on the round's tree, replay-path timing from the kernel's own stamps, rocprofv3
# evidence of the adopted K1 build, the default bench line, one tracer run of hw5 with 2-launch graphs.
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -5 $O/gputests.log
python3 bench/replay_stamps.py b200 b512 b1024 > $O/replay_stamps.txt 2>&1; cat $O/replay_stamps.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
bash bench/profile.sh r03 > $O/profile.log 2>&1; tail -22 $O/profile.log
bash bench/run_testcases.sh > $O/testcases_wall.txt 2>&1; tail -16 $O/testcases_wall.txt
NB_GRAPH_CHUNK=2 NB_HW5_CLEAN_EXIT=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/replay_b200 -o replay -- ./bin/hw5 tests/golden/testcases/b200.in /tmp/b200.out > $O/replay_b200.log 2>&1
echo "hw5 b200 with NB_GRAPH_CHUNK=2 under rocprofv3 --kernel-trace --stats: rc=$?" | tee $O/replay_b200_rc.txt
cmp /tmp/b200.out tests/golden/testcases/b200.out && echo "b200 output identical under the tracer" | tee -a $O/replay_b200_rc.txt
find $O/replay_b200 -name "*kernel_stats.csv" -exec cp {} $O/replay_b200_kernel_stats.csv \; 2>/dev/null
find $O -name "*.csv" -size +1M -delete
