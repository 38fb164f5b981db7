# bench/debug/r05_profile_run.sh — the one gpurun call behind profiles/r05_bench_default.json, r05_rocprof_summary.md, r05_create_timing.txt and the
# rehearsal lines (round 5); run from the repository root on the GPU box.
set -x
O=${1:-gpurun_out/r05d}; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
bench/profile.sh r05 > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
python bench/create_timing.py bench/ab/kahan/libnbody_amd.so nthu_ipc_nbody-simulation_amd/libnbody_amd.so > $O/create_timing.txt 2>&1 || { tail $O/create_timing.txt; exit 1; }
python bench.py --gpus 8 --exchange copy-one-gpu --steps 10 --warmup 2 > $O/rehearsal_native_8ranks_1gpu.json 2> $O/rehearsal_native_8ranks_1gpu.err; echo rc=$?
python bench.py --gpus 8 --steps 10 --warmup 2 > $O/ladder_native_8ranks_1gpu.json 2> $O/ladder_native_8ranks_1gpu.err; echo rc=$?
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 4 --backend gloo --single-device --steps 10 --warmup 2 > $O/rehearsal_torch_4ranks_1gpu.json 2> $O/rehearsal_torch_4ranks_1gpu.err; echo rc=$?
tail -3 $O/create_timing.txt
