#!/bin/bash
# bench/profile.sh <tag> [bench.py args...] — rocprofv3 evidence for the force kernel, run ON THE GPU BOX:
#   1. --kernel-trace --stats          -> per-kernel average duration (must agree with bench.py's HIP-event time)
#   2. --pmc FETCH_SIZE                -> HBM read bytes  (own pass: TCC slots; x2 gfx950 correction, see parser)
#   3. --pmc WRITE_SIZE                -> HBM write bytes (own pass)
#   4. --pmc SQ_* / GRBM_GUI_ACTIVE    -> VALU-busy evidence
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); bench/parse_profile.py condenses it into profiles/.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r01}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-parity-spot --no-live-pmc --no-diagnostics}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
python3 bench/parse_profile.py $OUT $TAG
