"""Stand-in for a leg child of `bench.py --gpus P` (tests/test_bench_ladder.py passes it as --leg-program): no GPU, no product
code — it only plays the ways a real leg can end, chosen by the leg's own arguments, so that the orchestrating parent's ladder
(bench.run_ladder, orchestrate_native, orchestrate_torch) is exercised on the CPU:
    shared pairs over rccl / nccl / gloo     dies at once with an RCCL-looking message (rank 1 only under a launcher: rank 0
                                            would then wait for ever — the parents must stop it early)
    ordered pairs over rccl                 hangs without a word (killed at the leg's time limit)
    ordered pairs over nccl / gloo          completes
    copy exchange, shared pairs             prints the line of its timed region, then hangs in its diagnostics
    copy exchange, ordered pairs            completes
    host-staged exchange                    completes
"""
import json
import os
import sys
import time

argv = sys.argv[1:]


def opt(name, default=None):
    return argv[argv.index(name) + 1] if name in argv else default


rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
exchange = opt("--exchange") or opt("--backend", "nccl")
ordered = "--ordered-pairs" in argv
steps, gpus = int(opt("--steps", "10")), int(opt("--gpus", "1"))
assert "--leg-child" in argv, argv


def line(stage):
    return json.dumps({"metric": "body-pair interactions/sec", "value": 1.0e12 + steps, "unit": "pairs/s", "n_gpus": gpus, "steps": steps,
                       "ms_per_step": 1.0, "host": "native" if "--host" in argv else "torch", "exchange": exchange,
                       "roofline": {"kernel": "stub", "frac": 0.5}, "parity_spot": {"ok": True}, "stage": stage,
                       "wall_s": {"process": 0.1, "timed_region": 0.01}, "stub_world": world})


if exchange in ("rccl", "nccl", "gloo") and not ordered:
    if world == 1 or rank == 1:
        print("ncclReduceScatter: unhandled system error (stub)", file=sys.stderr)
        sys.exit(3)
    time.sleep(3600)
if exchange == "rccl" and ordered:
    time.sleep(3600)
if exchange.startswith("copy") and not ordered:
    print(line("timed_region"), flush=True)
    time.sleep(3600)
if rank == 0:
    print(line("timed_region"), flush=True)
    print(line("complete"), flush=True)
