#!/bin/bash
# GPU session r03d: configs[4]'s workload on ONE GPU — N = 2^24, fp32 pair math / fp64 accumulate + fp64 masters — run
# sustained with progress reports and a checkpoint, then resumed from that checkpoint; the resumed run must end on the
# same bits as the uninterrupted one.  (61 s per step: 2.8e14 pairs.)
set -o pipefail
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r03d
mkdir -p $O
N=16777216
CK=/tmp/ck_n2e24.nbst
python3 bench.py --bodies $N --precision f32acc64 --steps 7 --warmup 0 --report-every 1 --time-box 1500 \
    --checkpoint $CK --checkpoint-every 4 --no-cpu-baseline --conservation --dump-rows $O/rows_uninterrupted.npz \
    > $O/sustained.json 2> $O/sustained.err || { tail -5 $O/sustained.err; exit 1; }
ls -l $CK >> $O/sustained.err
python3 bench.py --bodies $N --precision f32acc64 --steps 3 --warmup 0 --report-every 1 --resume $CK \
    --no-cpu-baseline --no-parity-spot --dump-rows $O/rows_resumed.npz > $O/resumed.json 2> $O/resumed.err || { tail -5 $O/resumed.err; exit 1; }
python3 - <<'PY' | tee $O/resume_check.txt
import numpy as np
a, b = np.load("gpurun_out/r03d/rows_uninterrupted.npz"), np.load("gpurun_out/r03d/rows_resumed.npz")
print("uninterrupted run ended at step", int(a["step"]), "- resumed run at step", int(b["step"]))
same = bool(int(a["step"]) == int(b["step"]) and np.array_equal(a["idx"], b["idx"]) and
            np.array_equal(a["q"].view(np.uint64), b["q"].view(np.uint64)) and
            np.array_equal(a["v"].view(np.uint64), b["v"].view(np.uint64)))
print("64 strided rows of the fp64 masters (q, v) bit-identical after checkpoint + resume:", same)
print("max |dq|, |dv|:", float(np.abs(a["q"] - b["q"]).max()), float(np.abs(a["v"] - b["v"]).max()))
PY
cat $O/sustained.err | grep -v amdgpu.ids
# configs[3]'s shape through the native multi-GPU host with all 8 ranks on this one GPU (copy exchange), overlapped: 2 steps
./bin/nbody_bench 4194304 2 0 f32 8 1 copy-one-gpu > $O/n2e22_p8_one_gpu.txt 2>&1; cat $O/n2e22_p8_one_gpu.txt
