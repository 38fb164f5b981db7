#!/bin/bash
# bench/shard_pairs_threshold.sh LIB — bench/shard_pairs_ab.py (one rank's K1 step against its K1s share) around the sharing threshold n^2 / P,
# with a library built with a low threshold (make LIB=bench/ab/share/libnbody_amd.so EXTRA=-DNB_SYM_SHARE_MIN=1e8 lib)
LIB=${1:-bench/ab/share/libnbody_amd.so}
for np in "32768 2" "40960 2" "49152 2" "32768 4" "49152 4" "65536 4" "81920 4" "65536 8" "98304 8" "131072 8"; do
  set -- $np
  python3 -c "
import sys
sys.argv = ['shard_pairs_ab.py', '$1', '$2', '0']
sys.path.insert(0, '.')
import nbody_amd
from nbody_amd import capi
capi.use_library('$LIB').__enter__()
__file__ = 'bench/shard_pairs_ab.py'
exec(open(__file__).read())
" 2>&1 | grep -E "^round 2|Error" || exit 1
done
