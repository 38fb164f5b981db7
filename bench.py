#!/usr/bin/env python3
"""bench.py — body-pair interactions/second of the all-pairs step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--bodies 1048576] [--precision f32|f32acc64]

A "step" is one run_step of the whole system (samples/nbody.cc:51-89): all-pairs force + kick + drift on
synthetic uniform-random bodies (nbody_amd.synthetic, seed 42) already resident in HBM.  N=1 runs BASELINE
configs[2] (N=2^20, fp32, 1 GPU).  With --gpus P > 1 the same N=2^20 bodies are sharded by index over P GPUs with one
in-place RCCL all-gather of positions per step, i.e. STRONG scaling of the metric's own N (--bodies 4194304 gives
configs[3]), through one of the two hosts of that scheme:
  * typed as is (`python3 bench.py --gpus P`, no launcher — the reference is one command on its GPUs too, hw5.cu:618):
    the C-ABI host nb_sharded_* — this ONE process drives the P GPUs, ncclCommInitAll + one in-place ncclAllGather per
    GPU per step (csrc/nbody_sharded.cpp); the line says "host": "native";
  * under `python -m torch.distributed.run --nproc-per-node P` (WORLD_SIZE = P): one process per GPU with
    torch.distributed for the collective (nbody_amd.distributed); "host": "torch".  Rank 0 then also runs the native
    host once as a bounded child process and embeds its line under "native_host".

Rank 0 prints ONE JSON line.  `value` = N(N-1)*K / wall (max over ranks, barrier + synchronize on both sides).
`roofline` prices the force kernel against the fp32 vector-FMA peak (157.3 TFLOP/s = the dense f32 MFMA peak in
MI355X_MICROARCH.md; the kernel is VALU/rsqrt work and deliberately does not use MFMA) at 20 flop per pair, from
the kernel's own duration measured with HIP events on the launch stream.  `cpu_baseline` times the reference's
run_step (oracle/_ref, compiled from samples/nbody.cc) — or the oracle port when that build is absent — on a
bounded sample of the same input, 1 thread, as the reference runs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 20            # GPU Gems 3 ch.31 convention
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
NOMINAL_CLOCK_HZ = 2.4e9       # the clock PEAK_FP32_TFLOPS is priced at
ISSUE_CEILING_FRAC = 0.62     # K1 (every ordered pair): what the pair loop's instruction mix can issue
ISSUE_CEILING_FRAC_SYM = 0.92   # K1s (every unordered pair once): 16 packed VALU + 2 v_rsq_f32 per 4 interactions + 14 DPP moves / 32


_JSON_FD = None
_T0 = time.perf_counter()  # process start, for the line's wall_s


def only_the_json_line_on_stdout():
    """Rank 0 prints ONE JSON line: libraries that write to the process's stdout on their own (librccl prints a banner — ROCm
    version, hostname, library path — when it is loaded) are sent to stderr at the file-descriptor level; emit() writes
    the line to the real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def with_wall(out):
    """What a clock around the whole command sees next to what the line's value is computed from."""
    out["wall_s"] = {"process": round(time.perf_counter() - _T0, 2),
                     "timed_region": round(out["ms_per_step"] * out["steps"] * 1e-3, 3),
                     "note": "process = imports, set-up, warm-up, the timed region (the K steps `value` and `ms_per_step` come "
                             "from) and the diagnostics after it (parity_spot, cpu_baseline, PMC passes, variants, ...)"}
    return out


def emit(line):
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_sample=16384, steps=2):
    """Reference run_step on the first n_sample synthetic bodies, single thread (what samples/nbody.cc is)."""
    import numpy as np
    from nbody_amd import synthetic
    from oracle import oracle as O
    q, v, m = synthetic.bodies(n_sample)
    s = O.System(n_sample)
    s.q[:], s.v[:], s.m[:] = q, v, m
    pairs = n_sample * (n_sample - 1) * steps
    if O.have_reference():
        kind = "reference"
        # the reference hard-codes dt=60, eps=1e-3, G=6.674e-11 (nbody.cc:10-13): same arithmetic per pair
        t0 = time.perf_counter()
        O.ref_run_steps(s, 1, steps)
        dt = time.perf_counter() - t0
        what = "oracle/_ref run_step (samples/nbody.cc compiled in place)"
    else:
        kind = "port"
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        p = O.make_params(dt=synthetic.DT, eps=synthetic.EPS, G=synthetic.G)
        t0 = time.perf_counter()
        O.run_steps(s, 1, steps, params=p, omp=False)
        dt = time.perf_counter() - t0
        what = "oracle/nbody_oracle.c run_step (bit-identical restatement)"
    assert np.isfinite(s.q).all()
    return {"value": pairs / dt, "unit": "pairs/s", "cores": 1, "kind": kind, "cpu_model": cpu_model(),
            "sample": f"{what}, {steps} steps on the first {n_sample} bodies of the same synthetic input "
                      f"({pairs:.3g} pairs, {dt:.1f} s)"}


def cpu_baseline_all_cores(n_total, rows=2048):
    """The oracle port (bit-identical restatement of run_step's accel phase), OpenMP over target rows on every host
    core, on `rows` strided targets against ALL n_total sources of the bench input."""
    import numpy as np
    from nbody_amd import synthetic
    from oracle import oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # the box's CPU share for one GPU
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the OpenMP build of the oracle is first loaded
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    q, _, m = synthetic.bodies(n_total)
    lo = (n_total // 2 // rows) * rows  # a contiguous block in the middle of the index range
    O.accel_rows(q, m, synthetic.G, synthetic.EPS, lo, lo + 8, omp=True)  # thread pool warm-up
    t0 = time.perf_counter()
    a = O.accel_rows(q, m, synthetic.G, synthetic.EPS, lo, lo + rows, omp=True)
    dt = time.perf_counter() - t0
    assert np.isfinite(a).all()
    pairs = rows * (n_total - 1)
    return {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"oracle/nbody_oracle.c accel rows (OpenMP, {cores} threads), {rows} targets x {n_total} sources "
                      f"of the same input ({pairs:.3g} pairs, {dt:.1f} s)"}


class PowerSampler:
    """Socket power and shader clock of ONE GPU during the timed region, read from the amdgpu hwmon files of the card with this PCI
    bus id (power1_input in microwatts, power1_cap, freq1_input in Hz) every 0.2 s by a host thread — file reads only, no GPU call,
    never raises.  The kernel is compute-bound and runs into the socket's power cap: this is the driver-visible form of
    profiles/r04b_power_and_clock.txt (rocm-smi by hand): what the chip drew and what clock it held while `value` was measured."""

    def __init__(self, pci_bus_id, period=0.2, root="/sys/class/drm"):
        import glob
        import threading
        self.dir, self.samples, self.period = None, [], period
        want = pci_bus_id.lower()
        for d in glob.glob(os.path.join(root, "card*", "device")):
            try:
                if os.path.realpath(d).lower().endswith(want):
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "power1_input")):
                        self.dir = hw[0]
                        break
            except OSError:
                pass
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.is_set():
            p, f = self._read("power1_input"), self._read("freq1_input")
            if p is not None:
                self.samples.append((p * 1e-6, (f or 0.0) * 1e-6))
            self._stop.wait(self.period)

    def start(self):
        if self._thread:
            self._thread.start()
        return self

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(timeout=2)
        if not self.samples:
            return None
        w = [a for a, _ in self.samples]
        mhz = [b for _, b in self.samples if b > 0]
        cap = self._read("power1_cap")
        return {"mean_w": sum(w) / len(w), "max_w": max(w), "cap_w": cap * 1e-6 if cap else None,
                "sclk_mhz_mean": sum(mhz) / len(mhz) if mhz else None, "sclk_mhz_min": min(mhz) if mhz else None,
                "samples": len(w), "source": "amdgpu hwmon (power1_input, freq1_input) of this GPU, every %.1f s during the timed region" % self.period}


def load_traffic(n_bodies, world, kernel, j_split):
    """HBM bytes per step of the force + reducer launches (None when the committed profile predates the reducer being
    counted), and the VALU-busy fraction, from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    bench/parse_profile.py on the builder's box — NOT measured by this run) — only when that profile was taken on THIS
    kernel, body count, rank count and source split; any other configuration reports null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except Exception:
        return None, None, None, None
    e = t.get(f"n{n_bodies}_p{world}", {})
    if e.get("kernel") != kernel or e.get("j_split") != j_split:
        return None, None, None, None
    return e.get("hbm_bytes_per_step"), e.get("reduce_share_of_span"), e.get("valu_busy"), e.get("tag")


def live_pmc(argv_tail, kernels=("nbody_force_f32", "nbody_reduce_update_f32"), timeout=120):
    """HBM traffic of one step's launches (force kernel + reducer — the same launches `kernel_ms` spans) and VALU-busy of
    the force kernel, measured BY THIS RUN: three short child runs of this same program
    (2 steps each) under `rocprofv3 --pmc`, one counter group per pass as the guide prescribes — FETCH_SIZE, WRITE_SIZE,
    then SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE — with the program directly after `--`.  FETCH_SIZE / WRITE_SIZE are KiB;
    on gfx950 FETCH_SIZE reports half of a streaming read and is doubled (MI355X_MICROARCH.md, HBM section).  Returns a
    dict, or None when the tool is missing or a pass fails (the committed profile is quoted then, and said to be)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3")
    if not tool:
        return None
    out = tempfile.mkdtemp(prefix="nb_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
             "--no-parity-spot", "--no-live-pmc"] + list(argv_tail)

    def one_pass(tag, counters):
        d = os.path.join(out, tag)
        cmd = [tool, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", tag, "--"] + child
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env, cwd="/tmp")
        if p.returncode != 0:
            raise RuntimeError(f"rocprofv3 pass {tag}: rc={p.returncode}")
        vals = {}  # kernel family -> counter -> values of its launches
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    for k in kernels:
                        if k in r.get("Kernel_Name", ""):
                            vals.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in vals.items()}

    try:
        f = one_pass("fetch", ["FETCH_SIZE"])
        w = one_pass("write", ["WRITE_SIZE"])
        q = one_pass("sq", ["SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"])
        force, red = kernels
        per = {}
        for k in kernels:  # one launch of each per step (plan: <= 64 slices); a step without slices has no reducer
            if k in f and k in w:
                per[k] = {"hbm_bytes": (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0,
                          "fetch_kib_raw": f[k]["FETCH_SIZE"], "write_kib": w[k]["WRITE_SIZE"]}
        res = {"hbm_bytes_per_step": sum(v["hbm_bytes"] for v in per.values()),
               "force": per.get(force), "reducer": per.get(red),
               "valu_busy": q[force]["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * q[force]["GRBM_GUI_ACTIVE"] / 8),
               "shader_cycles_per_launch": q[force]["GRBM_GUI_ACTIVE"] / 8}
    except Exception as e:  # noqa: BLE001  (tool crash, timeout, missing counter: fall back to the committed profile)
        res = {"error": f"{type(e).__name__}: {e}"}
    shutil.rmtree(out, ignore_errors=True)
    return res


def parity_spot(torch, sysm, n, acc64, compute_kw, rows=64):
    """The published number carries its own proof (SURVEY §8(d) "Parity on synthetic"): after the timed region, the
    accelerations of ALL targets on the positions the run ended on — one accel-only launch of the same kernel family with
    the same plan (register blocking, source slices, workgroup size) — and `rows` strided rows of it against the fp64
    oracle (reference arithmetic, samples/nbody.cc:56-74).  Error relative to sum_j |a_ij| (the net force on a uniform
    cloud cancels heavily); bound 1e-5 for fp32 sums, 1e-6 for fp64-accumulated ones."""
    import numpy as np
    from nbody_amd import capi, synthetic
    from oracle import oracle as O
    pos = sysm.positions
    rec = 32 if acc64 else 16
    acc = torch.empty((n, 4), dtype=torch.float64 if acc64 else torch.float32, device=pos.device)
    from nbody_amd.distributed import workspace_bytes
    ws = torch.empty(workspace_bytes(n, n, acc64, **compute_kw), dtype=torch.uint8, device=pos.device)
    capi.launch_f32(pos.data_ptr(), 0, n, 0, n, synthetic.EPS ** 2, synthetic.DT,
                    torch.cuda.current_stream(pos.device).cuda_stream, accel_only=True, acc_ptr=acc.data_ptr(),
                    acc64=acc64, workspace_ptr=ws.data_ptr(), workspace_bytes=ws.numel(), **compute_kw)
    torch.cuda.synchronize()
    assert acc.element_size() * 4 == rec
    idx = [(k * (n // rows) + (k * 37) % (n // rows)) % n for k in range(rows)]
    idx[0], idx[-1] = 0, n - 1
    a_gpu = acc[idx, :3].cpu().numpy().astype(np.float64).T
    p = pos.cpu().numpy().astype(np.float64)
    q, m = np.ascontiguousarray(p[:, :3].T), np.ascontiguousarray(p[:, 3] / synthetic.G)
    t0 = time.perf_counter()
    a, ab = O.accel_rows_at(q, m, synthetic.G, synthetic.EPS, idx, want_abs=True, omp=True)
    worst = float((np.abs(a_gpu - a).max(axis=0) / ab).max())
    tol = 1e-6 if acc64 else 1e-5
    return {"rows": rows, "pairs_checked": rows * (n - 1), "max_err_over_sum_abs": worst, "tol": tol,
            "ok": bool(worst < tol), "oracle_s": round(time.perf_counter() - t0, 2),
            "what": "accel-only launch (same plan as the timed step) on the final positions vs oracle/ fp64 rows"}


PEAK_FP64_TFLOPS = 78.6      # MI355X_MICROARCH.md: fp64 vector FMA, half the packed-fp32 rate


def other_precision_path(torch, n, device, mode, rows=64, steps=5):
    """N = 1, after the timed region: the same synthetic system in another arithmetic mode through the plain C-ABI context
    (nb_create / nb_set_state / nb_step_timed — what INTEGRATION.md binds), `steps` untimed-region steps timed with HIP events
    on the context's stream, then nb_accel on the state it ended on against `rows` oracle rows.
      f32acc64  BASELINE configs[4]'s arithmetic: fp32 pair math, fp64 sums, fp64 q,v masters (tolerance 1e-6 of sum|a_ij|)
      f64       the testcases' arithmetic at large n: K1s-f64, every unordered pair once in fp64 (tolerance 1e-12), priced
                against the 78.6 TFLOP/s fp64 vector peak."""
    import numpy as np
    from nbody_amd import capi, synthetic
    from oracle import oracle as O
    prec, tol, peak = {"f32acc64": (capi.NB_F32_ACC64, 1e-6, PEAK_FP32_TFLOPS), "f64": (capi.NB_F64, 1e-12, PEAK_FP64_TFLOPS)}[mode]
    q, v, m = synthetic.bodies(n)
    dev = device.index or 0
    with capi.Context(n, prec, dev, G=synthetic.G, eps=synthetic.EPS, dt=synthetic.DT) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 1)
        ms = ctx.step_timed(2, steps)
        kernel = ctx.kernel_name() if mode != "f64" else "nbody_force_sym_f64 (K1s-f64: every unordered pair once, fp64) + its reducer"
        q1, _ = ctx.get_state()
        a = ctx.accel(2 + steps)
    idx = [(k * (n // rows) + (k * 37) % (n // rows)) % n for k in range(rows)]
    idx[0], idx[-1] = 0, n - 1
    if mode == "f64":
        qs, ms_ = np.ascontiguousarray(q1), m
    else:  # the pair loop reads the fp32 copies of the fp64 masters, and G*m rounded once
        qs = np.ascontiguousarray(q1.astype(np.float32).astype(np.float64))
        ms_ = (synthetic.G * m).astype(np.float32).astype(np.float64) / synthetic.G
    t0 = time.perf_counter()
    ref, ab = O.accel_rows_at(qs, ms_, synthetic.G, synthetic.EPS, idx, want_abs=True, omp=True)
    worst = float((np.abs(a[:, idx] - ref).max(axis=0) / ab).max())
    return {"ms_per_step": ms, "frac": FLOP_PER_PAIR * n * (n - 1) / (ms * 1e-3) / 1e12 / peak, "peak": peak,
            "pairs_per_s": n * (n - 1) / (ms * 1e-3), "bodies": n, "kernel": kernel, "steps": steps, "dtype": mode,
            "parity": {"rows": rows, "max_err_over_sum_abs": worst, "tol": tol, "ok": bool(worst < tol),
                       "oracle_s": round(time.perf_counter() - t0, 2)},
            "what": ("nb_step_timed of a C-ABI context on the same synthetic bodies (HIP events on the context's stream), then "
                     "nb_accel of the final state vs oracle/ fp64 rows")}


def sharded_check(torch, dist, world, rank, device, dev_index, backend):
    """world > 1 only, after the timed region: the same sharded stepper (index shards, tgt_off != 0 launches, in-place
    RCCL all-gather, ping-pong) on a small system, against the UNSHARDED run of the same bodies on rank 0's GPU through
    nb_step.  Proves on the real multi-GPU node what tests/test_gpu_distributed.py rehearses on one GPU."""
    import numpy as np
    from nbody_amd import capi, synthetic
    from nbody_amd.distributed import ShardedSystem, shard_range
    n, steps, dt = 32768, 3, 1e-2
    lo, hi = shard_range(n, rank, world)
    pos, vel = synthetic.body4_f32(n, lo, hi)
    s = ShardedSystem(n, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, dt, device)
    for _ in range(steps):
        s.step()
    torch.cuda.synchronize()
    bits = s.positions.contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)  # checksum of the gathered array
    if backend != "nccl":
        bits = bits.cpu()
    hi_, lo_ = bits.clone(), bits.clone()
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    agree = bool((hi_ == lo_).item())
    out = None
    if rank == 0:
        q, v, m = synthetic.bodies(n)
        with capi.Context(n, capi.NB_F32, dev_index, G=synthetic.G, eps=synthetic.EPS, dt=dt) as ctx:
            ctx.set_state(q, v, m)
            ctx.step(1, steps)
            q1, _ = ctx.get_state()
        diff = float(np.abs(s.positions[:, :3].cpu().numpy().T.astype(np.float64) - q1).max())
        moved = float(np.abs(q1 - q.astype(np.float32)).max())
        out = {"bodies": n, "steps": steps, "ranks_hold_identical_positions": agree,
               "max_abs_diff_vs_unsharded": diff, "max_displacement": moved,
               "ok": bool(agree and diff < 5e-7 and moved > 1e-6), "exchange": s.exchange_mode}
    return out


def spot_rows(n, world, rows=64):
    """`rows` target indices spread over EVERY rank's shard (the first and the last body of the system among them)."""
    per = n // world
    k_per = max(1, rows // world)
    stride = max(1, per // k_per)
    idx = [r * per + (k * stride + (k * 37) % stride) % per for r in range(world) for k in range(k_per)]
    idx[0], idx[-1] = 0, n - 1
    return idx


def step_rows_vs_oracle(q0, gm, idx, v0_rows, v1_rows, acc64, world):
    """One run_step of the sharded system against the oracle (samples/nbody.cc:56-88), on the rows `idx`:
    a = (v' - v) / dt recovered from the velocities before and after ONE more step, against oracle.accel_rows on the
    positions the step started from (q0: (3, N) — the fp32 records the pair loop reads, widened; gm: G*m as stored).
    Error relative to sum_j |a_ij|; v' stored in fp32 recovers a only to ulp(v')/dt, which is allowed on top of the
    tolerance (and reported).  Bound 1e-5 for fp32 sums, 1e-6 for fp64-accumulated ones."""
    import numpy as np
    from nbody_amd import synthetic
    from oracle import oracle as O
    dt = float(np.float32(synthetic.DT))  # the kernels step with the fp32 value of dt
    m = np.ascontiguousarray(gm / synthetic.G)
    tol = 1e-6 if acc64 else 1e-5
    worst = worst_slack = 0.0
    ok = True
    t0 = time.perf_counter()
    a_all, ab_all = O.accel_rows_at(q0, m, synthetic.G, synthetic.EPS, idx, want_abs=True, omp=True)
    for c in range(len(idx)):
        a, ab = a_all[:, c:c + 1], ab_all[c:c + 1]
        a_gpu = (v1_rows[:, c] - v0_rows[:, c]) / dt
        slack = 0.0 if acc64 else 2.0 ** -23 * float(np.abs(v1_rows[:, c]).max()) / dt
        err = float(np.abs(a_gpu - a[:, 0]).max())
        ok = ok and err <= tol * ab[0] + slack
        worst, worst_slack = max(worst, err / ab[0]), max(worst_slack, slack / ab[0])
    per = len(q0[0]) // world
    return {"rows": len(idx), "rows_per_rank": len(idx) // world, "ranks_covered": len({i // per for i in idx}),
            "pairs_checked": len(idx) * (len(q0[0]) - 1), "max_err_over_sum_abs": worst, "tol": tol,
            "fp32_velocity_recovery_slack_over_sum_abs": worst_slack, "ok": bool(ok),
            "oracle_s": round(time.perf_counter() - t0, 2),
            "what": "one more sharded step after the timed region; (v' - v)/dt of rows from every rank's shard vs "
                    "oracle/ fp64 accelerations on the positions that step started from"}


_LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                 "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS")


def child_env(**extra):
    """Environment for a child process that must not think it is a rank of this job."""
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCHER_ENV and not k.startswith("TORCHELASTIC_")}
    env.update(extra)
    return env


def run_process(cmd, timeout, env=None, should_abort=None):
    """A bounded child in its own session: -> (rc | None on timeout or abort, stdout, stderr).  On timeout the whole process
    group is killed and what the child had written so far is still returned (a leg prints its line as soon as the timed
    region closes, so diagnostics that hang afterwards cost the diagnostics, not the measurement).  `should_abort()` is
    polled twice a second: the ranks of one leg stop together when one of them has failed."""
    import signal
    import subprocess
    import tempfile
    with tempfile.TemporaryFile(mode="w+", dir="/tmp") as fo, tempfile.TemporaryFile(mode="w+", dir="/tmp") as fe:
        p = subprocess.Popen(cmd, stdout=fo, stderr=fe, text=True, env=env if env is not None else child_env(), cwd=ROOT,
                             start_new_session=True)
        t0 = time.perf_counter()
        rc, note = None, ""
        while True:
            rc = p.poll()
            if rc is not None:
                break
            if time.perf_counter() - t0 > timeout:
                note = ""
                break
            if should_abort is not None and should_abort():
                note = "[stopped: another rank of this leg failed]\n"
                break
            time.sleep(0.05 if time.perf_counter() - t0 < 2 else 0.5)
        if rc is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # the exact group this call started
            except OSError:
                pass
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                pass
        fo.seek(0)
        fe.seek(0)
        return rc, fo.read(), note + fe.read()


def last_json_line(text):
    for ln in reversed([ln for ln in text.splitlines() if ln.startswith("{")]):
        try:
            return json.loads(ln)
        except ValueError:
            continue
    return None


def leg_record(name, rc, out, err, timeout, seconds, what=""):
    """What one leg left behind: its JSON line (the last complete one) and/or how it ended."""
    line = last_json_line(out)
    rec = {"name": name, "seconds": round(seconds, 1)}
    if what:
        rec["what"] = what
    if rc is None and "[stopped: another rank of this leg failed]" in err:
        rec["error"] = "stopped: another rank of this leg failed"
    elif rc is None:
        rec["timeout"] = f"killed after {timeout:.0f} s"
    elif rc != 0:
        rec["error"] = f"rc={rc}"
    if rc != 0 or line is None:
        tail = [ln for ln in err.strip().splitlines() if ln.strip()][-6:]
        rec["stderr_tail"] = " | ".join(tail)[-600:]
        if line is None and "error" not in rec and "timeout" not in rec:
            rec["error"] = "no JSON line"
    rec["ok"] = line is not None and "value" in line  # a line from the timed region counts even if the diagnostics died
    if rec["ok"] and (rc != 0 or line.get("stage") != "complete"):
        rec["diagnostics_incomplete"] = True
    return rec, line


def run_ladder(legs, runner, budget_s, leg_timeout_s, log=None, sync=None):
    """The measured step of a multi-GPU line, as a LADDER of fresh child processes: legs in order until one yields a line;
    a leg that fails or hangs is recorded ({name, error | timeout, stderr_tail}) and the next form of the same step is tried
    — shared pairs over RCCL, then north_star's literal scheme (ordered pairs, all-gather only), then the copy-engine
    exchange.  `runner(leg, timeout) -> (rc | None, stdout, stderr)`.  -> (records, index of the leg whose line counts | None,
    that line | None).  Never raises for a failing leg: the first real 8-GPU contact must come back with a JSON line.
    `sync(x) -> x of rank 0`: when several parent processes walk the ladder together (one per rank under torch.distributed.run)
    every decision that depends on a clock is rank 0's."""
    t0 = time.perf_counter()
    records, chosen, line = [], None, None
    for i, leg in enumerate(legs):
        if leg.get("skip"):
            records.append({"name": leg["name"], "skipped": leg["skip"], "ok": False})
            continue
        left = budget_s - (time.perf_counter() - t0)
        if sync:
            left = sync(left)
        if left < min(20.0, leg_timeout_s):
            records.append({"name": leg["name"], "skipped": "out of time budget", "ok": False})
            continue
        # a leg may use what is left minus a reserve of 90 s for every leg still behind it (two RCCL legs that hang until
        # their limit must not starve the copy-engine legs), but always at least a minute
        behind = sum(1 for lg in legs[i + 1:] if not lg.get("skip"))
        timeout = min(leg_timeout_s, max(min(60.0, left), left - 90.0 * behind))
        t1 = time.perf_counter()
        if log:
            log(f"leg {i} {leg['name']}: starting (limit {timeout:.0f} s)")
        try:
            rc, out, err = runner(leg, timeout)
        except Exception as e:  # noqa: BLE001  (could not even start the child)
            rc, out, err = -1, "", f"{type(e).__name__}: {e}"
        rec, ln = leg_record(leg["name"], rc, out, err, timeout, time.perf_counter() - t1, leg.get("what", ""))
        records.append(rec)
        if log:
            log(f"leg {i} {leg['name']}: " + ("ok" if rec["ok"] else rec.get("error") or rec.get("timeout") or "failed"))
        if rec["ok"]:
            chosen, line = i, ln
            break
    return records, chosen, line


def summarise_leg_line(r):
    """The part of another leg's line worth keeping beside the one that counts."""
    if not r:
        return None
    keep = {k: r[k] for k in ("value", "ms_per_step", "steps", "kernel_ms_per_rank", "non_kernel_ms_per_step", "exchange", "overlap",
                              "host", "pairs") if k in r}
    if "roofline" in r:
        keep["kernel"], keep["roofline_frac"] = r["roofline"].get("kernel"), r["roofline"].get("frac")
    if "parity_spot" in r:
        keep["parity_spot_ok"] = r["parity_spot"].get("ok")
    return keep


def probe_device_count(timeout=120):
    """GPUs this box shows, asked of a throw-away child (the orchestrating parent never initialises HIP)."""
    rc, out, _ = run_process([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import nbody_amd; "
                              "from nbody_amd import capi; print(capi.device_count())" % ROOT], timeout)
    try:
        return int(out.strip().splitlines()[-1]) if rc == 0 else 0
    except (ValueError, IndexError):
        return 0


def replicas_check(devices, cases=("b1024", "b200"), timeout=60):
    """The reference's OWN multi-GPU mode on these GPUs (hw5.cu:564-567,587-588: task parallelism — P1, P2 and the
    Problem-3 runs spread over the devices, P2's arrival snapshot crossing from one GPU to another, hw5.cu:482-484):
    `NB_DEVICES=d0,d1 bin/hw5 bN.in out` as a child, wall time and the three output lines against the golden file."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "bin", "hw5")
    devs = ",".join(str(d) for d in devices[:2])
    out = {"devices": devs, "what": "NB_DEVICES=%s bin/hw5 (whole reference program, scenarios spread over the GPUs)" % devs}
    for case in cases:
        inp = os.path.join(ROOT, "tests", "golden", "testcases", case + ".in")
        gold = os.path.join(ROOT, "tests", "golden", "testcases", case + ".out")
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            dst = os.path.join(d, case + ".out")
            t0 = time.perf_counter()
            try:
                p = subprocess.run([exe, inp, dst], env=child_env(NB_DEVICES=devs), stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, text=True, timeout=timeout)
                wall = time.perf_counter() - t0
                same = p.returncode == 0 and open(dst, "rb").read() == open(gold, "rb").read()
                out[case] = {"wall_s": round(wall, 3), "rc": p.returncode, "byte_identical": bool(same)}
                if p.returncode != 0:
                    out[case]["stderr_tail"] = p.stderr[-300:]
            except subprocess.TimeoutExpired:
                out[case] = {"error": f"timeout after {timeout} s"}
            except OSError as e:
                out[case] = {"error": f"{type(e).__name__}: {e}"}
    return out


def step_kernels(kname):
    """(force kernel family, reducer family) of a step, for matching rocprofv3 rows."""
    return ("nbody_force_sym_f32", "nbody_reduce_sym_f32") if kname.startswith("nbody_force_sym_f32") else \
        ("nbody_force_f32", "nbody_reduce_update_f32")


def roofline_block(achieved, kname, k_ms, tpl, jsp, wgs, acc64, traffic=None, traffic_source=None, live=None,
                   valu_busy=None, reduce_share=None, shared_pairs=False):
    sym = kname.startswith("nbody_force_sym_f32")
    reducer = (f"nbody_reduce_sym_f32<{'true' if acc64 else 'false'}, {2 if shared_pairs else 0}>" if sym else
               f"nbody_reduce_update_f32<{'true' if acc64 else 'false'}, false>")
    ceiling = ISSUE_CEILING_FRAC_SYM if sym else ISSUE_CEILING_FRAC
    detail = ("every UNORDERED pair once (Newton's third law; `value` counts the N(N-1) ordered interactions the reference "
              "evaluates, nbody.cc:57-60): per 4 interactions x 64 lanes a SIMD issues 16 packed fp32 VALU ops (4 cycles each) "
              "+ 2 v_rsq_f32 (8 cycles each) = 80 cycles, plus 14 v_mov_b32_dpp (4 cycles each, measured) per 8 such sets that "
              "rotate the travelling sources -> 696 cycles per 2048 interactions = 21.75 per 64 = 0.92 of peak at 20 flop/pair "
              "(nbody_kernels_f32_sym.hip, profiles/r01_ubench_valu_rate.txt, r04_ubench_dpp_rate.txt)") if sym else (
              "per 64 pairs a SIMD issues 12 fp32 VALU ops (2 cycles each, packed: 6 x 4) "
              "+ 1 v_rsq_f32 (8 cycles, does not overlap VALU) = 32 cycles -> 1024 SIMDs "
              "x 2.4 GHz x 64/32 = 4.9e12 pairs/s = 0.62 of peak at 20 flop/pair "
              "(profiles/r01_ubench_valu_rate.txt)")
    return {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
            "traffic_detail": ({"force_kernel": live["force"]["hbm_bytes"] if live.get("force") else None,
                                "reducer": live["reducer"]["hbm_bytes"] if live.get("reducer") else None,
                                "algorithmic": None} if live and "error" not in live else None),
            "traffic_source": traffic_source,
            "live_pmc": live,
            "valu_busy": valu_busy,
            # the clock the chip held in the force kernel (cycles of the PMC pass / the timed region's kernel time): the
            # peak is priced at the nominal 2.4 GHz, the socket's power cap decides what is held (DESIGN.md, K1s ceiling)
            "clock_ghz_held": (live["shader_cycles_per_launch"] / (k_ms * 1e-3) / 1e9
                               if live and live.get("shader_cycles_per_launch") else None),
            "frac_at_clock_held": (achieved / PEAK_FP32_TFLOPS / (live["shader_cycles_per_launch"] / (k_ms * 1e-3) / NOMINAL_CLOCK_HZ)
                                   if live and live.get("shader_cycles_per_launch") else None),
            "kernel": kname, "kernel_ms": k_ms,
            "kernel_ms_spans": [kname] + ([reducer] if (jsp > 1 or sym) else []),
            "pair_evaluation": "each unordered pair once, both bodies served" if sym else "every ordered pair",
            "reduce_share_of_span": reduce_share,
            "targets_per_lane": tpl, "j_split": jsp, "wg_size": wgs, "flop_per_pair": FLOP_PER_PAIR,
            # flops the hardware executes per COUNTED (ordered) interaction: K1s serves two of them with one evaluation of
            # 3 sub + 6 r2 + 1 rsq + 2 cube + 2 scale + 12 accumulate = 26 -> 13; K1 executes the convention's 20 itself
            # (3 + 6 + 1 rsq + 3 + 6 = 19 ops, counted as 20 like everyone since GPU Gems 3).  `frac` uses flop_per_pair
            "flop_per_pair_executed": 13 if sym else 20,
            "frac_executed": achieved / PEAK_FP32_TFLOPS * (13 if sym else 20) / FLOP_PER_PAIR,
            "issue_ceiling_frac": ceiling,
            "frac_of_issue_ceiling": achieved / PEAK_FP32_TFLOPS / ceiling,
            "issue_ceiling_detail": detail,
            "bound_detail": "compute-bound on the fp32 vector-FMA (VALU) pipe: peak 157.3 TFLOP/s (numerically "
                            "the dense f32 MFMA peak, which is why the contract's hbm|mfma enum would say "
                            "'mfma'); the kernel issues v_pk_*_f32 + v_rsq_f32 and no MFMA instruction",
            "valu_busy_detail": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), "
                                "same source as `traffic` (see traffic_source)"}


def workload_config(n, world, how):
    return {"workload": f"N={n} synthetic uniform-random bodies (splitmix64 seed 42), all-pairs "
                        f"force + fused kick-drift, eps=1e-3, dt=1e-4", "bodies": n,
            "parallelism": how if world > 1 else "single GPU"}


NATIVE_EXCHANGES = ("rccl", "copy", "copy-one-gpu", "host", "host-one-gpu")


def native_parallelism(P, sym, exchange):
    """config.parallelism of a native-host line, from what actually ran."""
    coll = {"host": ("-", "host-staged all-gather (D2H of the own slot into one pinned array, H2D of the others: no peer-to-peer)"),
            "host-one-gpu": ("-", "host-staged all-gather (all ranks on GPU 0: rehearsal)"),
            "rccl": ("ncclReduceScatter", "in-place ncclAllGather"),
            "copy": ("copy-engine reduce-scatter (hipMemcpyPeerAsync pulls + ordered sum)", "copy-engine all-gather (hipMemcpyPeerAsync)"),
            "copy-one-gpu": ("same-device copy reduce-scatter (all ranks on GPU 0: rehearsal)",
                             "same-device copy all-gather (all ranks on GPU 0: rehearsal)")}[exchange]
    return (f"index-sharded x{P}, one process, "
            + (f"unordered pairs shared by the GPUs, 1 {coll[0]} of partial forces + " if sym else "")
            + f"1 {coll[1]} of float4 positions/step")


def main_native(args):
    """The C-ABI host: ONE process drives the P GPUs through nb_sharded_* (csrc/nbody_sharded.cpp): per step and GPU its share
    of the unordered pairs (K1s) + ONE ncclReduceScatter of partial forces + kick-drift — or, with --ordered-pairs, every
    ordered pair of its own targets (K1) — and ONE in-place ncclAllGather(sendbuff = recvbuff + r*4N/P) of the positions:
    RCCL over xGMI, ncclCommInitAll, no torch, no launcher.  --exchange copy = both collectives as peer copies on the copy
    engines; copy-one-gpu = the same with every rank on device 0 (rehearsal of the whole P > 1 host on a one-GPU box; the ranks
    then share the chip).  Runs as a LEG of `bench.py --gpus P` (a bounded child of the orchestrating parent, see
    orchestrate_native), or directly with --host native.  Every wait is bounded (nb_sharded_set_deadline): a collective that
    never completes ends this process with a message and rc != 0 instead of hanging it."""
    import numpy as np
    import nbody_amd  # noqa: F401
    from nbody_amd import capi, synthetic
    if args.lib:
        capi.library_path = lambda: os.path.abspath(args.lib)
    P, n = args.gpus, args.bodies
    acc64 = args.precision == "f32acc64"
    exchange = args.exchange or "rccl"
    if exchange not in NATIVE_EXCHANGES:
        raise SystemExit(f"--exchange {exchange}: the native host takes {', '.join(NATIVE_EXCHANGES)} "
                         f"(in_place/staged/ring belong to the torch host: launch with torch.distributed.run)")
    if args.conservation or args.dump_rows:
        raise SystemExit("--conservation/--dump-rows: single rank or the torch host")
    devices = [0] * P if exchange.endswith("one-gpu") else list(range(P))
    lib_exchange = exchange.split("-")[0]  # rccl | copy | host
    prec = capi.NB_F32_ACC64 if acc64 else capi.NB_F32
    kw = dict(G=synthetic.G, eps=synthetic.EPS, dt=synthetic.DT)
    try:
        sh = capi.Sharded(n, devices, prec, overlap=args.overlap, exchange=lib_exchange,
                          ordered_pairs=args.ordered_pairs, deadline=args.deadline, **kw)
    except capi.NBodyError as e:  # no GPU (NB_ERR_NO_DEVICE), fewer than P GPUs, RCCL missing: fail loudly, no fallback
        raise SystemExit(f"bench.py --gpus {P} (native host, devices {devices}): {e}")
    q, v, m = synthetic.bodies(n)
    first_step = 0
    if args.resume:  # a checkpoint of this very run (nb_sharded_load_state refuses another n / precision / G / eps / dt)
        try:
            first_step = sh.load_state(args.resume)
        except capi.NBodyError as e:
            raise SystemExit(f"--resume {args.resume}: {e}")
    else:
        sh.set_state(q, v, m)
    del q, v
    info = sh.info()
    kname = sh.kernel_name()
    sym = kname.startswith("nbody_force_sym_f32")
    ranks = [sh.rank_info(r) for r in range(P)]
    if args.warmup:
        sh.step(args.warmup)
    # ---- timed region: K steps, all GPUs idle on both sides (nb_sharded_step_profiled synchronises every stream before
    #      its first launch and after its last; the per-rank event pairs are created before its clock starts)
    chunk = args.report_every if args.report_every else 1024
    every = args.checkpoint_every if (args.checkpoint and args.checkpoint_every) else 0
    kern = [0.0] * P
    steps_done = 0
    ckpt_s = []
    samplers = {}  # one per distinct GPU of the run: what each socket drew and the clock it held during the timed region
    for r in ranks:
        if r["pci_bus_id"] not in samplers:
            try:
                samplers[r["pci_bus_id"]] = PowerSampler(r["pci_bus_id"]).start()
            except Exception:  # noqa: BLE001  (evidence, never a reason to fail the measurement)
                pass
    t0 = time.perf_counter()
    while steps_done < args.steps:
        k = min(chunk, args.steps - steps_done, 1024)
        if every:
            k = min(k, every - steps_done % every)
        _, kms = sh.step_profiled(k)
        kern = [a + b * k for a, b in zip(kern, kms)]
        steps_done += k
        if every and steps_done % every == 0:  # (inside the timed region, like the torch host's)
            tc = time.perf_counter()
            sh.save_state(args.checkpoint, first_step + args.warmup + steps_done)
            ckpt_s.append(time.perf_counter() - tc)
        el = time.perf_counter() - t0
        if args.report_every and steps_done < args.steps:
            print(f"[bench] {steps_done}/{args.steps} steps, {el:.1f} s, "
                  f"{n * (n - 1) * steps_done / el:.4e} pairs/s sustained", file=sys.stderr, flush=True)
            if args.time_box and el > args.time_box:
                break
    wall = time.perf_counter() - t0
    power = {bus: smp.stop() for bus, smp in samplers.items()}
    kern = [x / steps_done for x in kern]
    per = info["targets_per_device"]
    k_ms = max(kern)  # the slowest rank prices the roofline
    achieved = FLOP_PER_PAIR * per * (n - 1) / (k_ms * 1e-3) / 1e12
    out = {
        "metric": "body-pair interactions/sec",
        "value": n * (n - 1) * steps_done / wall,
        "unit": "pairs/s",
        "n_gpus": P,
        "steps": steps_done,
        "warmup": args.warmup,
        "ms_per_step": wall / steps_done * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if not acc64 else "f32 pair math / f64 accumulate",
        "data": "synthetic",
        "config": workload_config(n, P, native_parallelism(P, sym, exchange)),
        "host": "native",
        "host_detail": "nb_sharded_* (csrc/nbody_sharded.cpp): ONE process, one stream per GPU, "
                       + ("ncclCommInitAll; per GPU and step " + ("one ncclReduceScatter of partial forces + " if sym else "")
                          + "one in-place ncclAllGather (RCCL over xGMI)" if exchange == "rccl"
                          else "the all-gather through a pinned host array, per-device copies only (NB_SHARDED_HOST_EXCHANGE)" if lib_exchange == "host"
                          else "P-1 peer copies per GPU per step and collective on the copy engines (NB_SHARDED_COPY_EXCHANGE)")
                       + f"; every wait bounded at {args.deadline:g} s per step (nb_sharded_set_deadline)",
        "exchange": exchange,
        "overlap": bool(args.overlap),
        "pairs": ("every unordered pair once, shared by the GPUs (K1s); every GPU's partial force on all bodies is "
                  "reduce-scattered to the shard owners") if sym else "every ordered pair of a GPU's own targets (K1)",
        "ranks": {"count": P, "distinct_devices": len({r["uuid"] for r in ranks}),
                  "exchange": ranks[0]["exchange"],
                  "comm_ranks_seen_by_every_rank": sorted({r["comm_ranks"] for r in ranks}),
                  "per_rank": [{k: r[k] for k in ("rank", "device", "comm_rank", "comm_device", "pci_bus_id", "uuid",
                                                  "name", "compute_units", "first_target", "targets")} for r in ranks]},
        "roofline": roofline_block(achieved, kname, k_ms, info["targets_per_lane"], info["j_split"], info["wg_size"], acc64,
                                   shared_pairs=sym and P > 1),
        "kernel_ms_per_rank": kern,
        "non_kernel_ms_per_step": wall / steps_done * 1e3 - k_ms,
    }
    if any(power.values()):
        out["roofline"]["power_per_gpu"] = {bus: ({k: pw[k] for k in ("mean_w", "max_w", "cap_w", "sclk_mhz_mean", "sclk_mhz_min", "samples")}
                                                  if pw else None) for bus, pw in power.items()}
    if sh.note:
        out["shared_pairs_note"] = sh.note
    if first_step:
        out["resumed_from_step"] = first_step
    if ckpt_s:
        out["checkpoints"] = {"count": len(ckpt_s), "seconds_each": [round(x, 2) for x in ckpt_s],
                              "bytes": os.path.getsize(args.checkpoint), "inside_timed_region": True}
    if exchange.endswith("one-gpu"):
        out["rehearsal"] = f"all {P} ranks on GPU 0 (they share the chip): the P > 1 host logic, not a {P}-GPU measurement"
    out["roofline"]["kernel_ms_detail"] = ("slowest rank's mean per step, HIP events on each rank's own compute stream "
                                           "around its launch sequence (nb_sharded_step_profiled); all ranks in kernel_ms_per_rank")
    if args.leg_child:  # the measurement is safe with the parent from here on, whatever the checks below do
        out["stage"] = "timed_region"
        emit(json.dumps(with_wall(out)))
    # ---- untimed: the line carries its own proof
    if not args.no_parity_spot:
        try:
            idx = spot_rows(n, P)
            q0, v0 = sh.get_state()
            sh.step(1)
            _, v1 = sh.get_state()
            q32 = np.ascontiguousarray(q0.astype(np.float32).astype(np.float64))  # what the pair loop reads
            gm = (synthetic.G * m).astype(np.float32).astype(np.float64)
            out["parity_spot"] = step_rows_vs_oracle(q32, gm, idx, v0[:, idx], v1[:, idx], acc64, P)
            del q0, v0, v1, q32
        except Exception as e:  # noqa: BLE001
            out["parity_spot"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    try:
        sh.close()
    except Exception as e:  # noqa: BLE001
        out.setdefault("diagnostics_errors", []).append(f"close: {type(e).__name__}: {e}")
    if not args.no_diagnostics:
        # agreement with the unsharded stepper on a small system (the oracle check is parity_spot above)
        try:
            nn, steps, dt = 32768, 3, 1e-2
            q, v, mm = synthetic.bodies(nn)
            with capi.Sharded(nn, devices, capi.NB_F32, G=synthetic.G, eps=synthetic.EPS, dt=dt, overlap=args.overlap,
                              exchange=lib_exchange, deadline=args.deadline) as s2:
                s2.set_state(q, v, mm)
                s2.step(steps)
                q2, _ = s2.get_state()
            with capi.Context(nn, capi.NB_F32, devices[0], G=synthetic.G, eps=synthetic.EPS, dt=dt) as ctx:
                ctx.set_state(q, v, mm)
                ctx.step(1, steps)
                q1, _ = ctx.get_state()
            diff, moved = float(np.abs(q2 - q1).max()), float(np.abs(q1 - q.astype(np.float32)).max())
            out["sharded_check"] = {"bodies": nn, "steps": steps, "max_abs_diff_vs_unsharded": diff,
                                    "max_displacement": moved, "ok": bool(diff < 5e-7 and moved > 1e-6)}
        except Exception as e:  # noqa: BLE001
            out["sharded_check"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    out["stage"] = "complete"
    emit(json.dumps(with_wall(out)))


# ---------------------------------------------------------------- `--gpus P`: the measured step as a ladder of bounded children

def passthrough_args(args, steps=None, diagnostics=True, parity=None):
    """What every leg child of this run inherits."""
    parity = diagnostics if parity is None else parity
    a = ["--gpus", str(args.gpus), "--bodies", str(args.bodies), "--precision", args.precision,
         "--steps", str(steps if steps is not None else args.steps), "--warmup", str(args.warmup if steps is None else 1),
         "--deadline", str(args.deadline), "--leg-child"]
    if args.lib:
        a += ["--lib", args.lib]
    if steps is None:
        if args.report_every:
            a += ["--report-every", str(args.report_every)]
        if args.time_box:
            a += ["--time-box", str(args.time_box)]
        if args.checkpoint:
            a += ["--checkpoint", args.checkpoint, "--checkpoint-every", str(args.checkpoint_every)]
        if args.resume:
            a += ["--resume", args.resume]
    if args.no_parity_spot or not parity:
        a += ["--no-parity-spot"]
    if args.no_diagnostics or not diagnostics:
        a += ["--no-diagnostics"]
    return a


def native_legs(args, ndev):
    """The ladder of the native host, the request first: (exchange, ordered pairs).  Default order: shared pairs over RCCL
    (reduce-scatter + all-gather), north_star's literal scheme (ordered pairs, ONE RCCL all-gather), the copy-engine exchange
    (no RCCL at all: shared, then ordered).  A box with fewer than P GPUs can only rehearse: the copy exchange with every rank
    on GPU 0 (said in the line: `rehearsal`)."""
    P = args.gpus
    req_ex = args.exchange or "rccl"
    if req_ex not in NATIVE_EXCHANGES:
        raise SystemExit(f"--exchange {req_ex}: the native host takes {', '.join(NATIVE_EXCHANGES)} "
                         f"(in_place/staged/ring belong to the torch host: launch with torch.distributed.run)")
    if req_ex.startswith("host") and args.overlap:
        raise SystemExit("--exchange host: the host-staged exchange is not overlapped")
    req = (req_ex, bool(args.ordered_pairs or args.overlap or req_ex.startswith("host")))
    order = [req]
    usable = lambda x: not (args.overlap and (not x[1] or x[0].startswith("host")))  # noqa: E731  (overlap = ordered pairs, not host-staged)
    if not req_ex.endswith("one-gpu"):
        order += [x for x in (("rccl", False), ("rccl", True), ("copy", False), ("copy", True), ("host", True)) if x != req and usable(x)]
    if ndev < P or req_ex.endswith("one-gpu"):
        order += [x for x in (("copy-one-gpu", False), ("copy-one-gpu", True), ("host-one-gpu", True)) if x not in order and usable(x)]
    legs = []
    for k, (ex, ordered) in enumerate(order):
        name = ("ordered_pairs" if ordered else "shared_pairs") + "_" + ex.replace("-", "_") + ("_overlap" if args.overlap else "")
        argv = ["--host", "native", "--exchange", ex] + (["--ordered-pairs"] if ordered else []) + (["--overlap"] if args.overlap else [])
        what = {"rccl": "RCCL: " + ("in-place all-gather of positions only (north_star's scheme)" if ordered else
                                    "reduce-scatter of partial forces + in-place all-gather of positions"),
                "copy": "copy engines (hipMemcpyPeerAsync), no RCCL", "copy-one-gpu": f"all {P} ranks on GPU 0 (rehearsal)",
                "host": "all-gather through a pinned host array: per-device copies only, no peer-to-peer, no RCCL (last resort)",
                "host-one-gpu": f"host-staged all-gather, all {P} ranks on GPU 0 (rehearsal)"}[ex]
        leg = {"name": name, "argv": argv, "what": ("every ordered pair of a GPU's own targets (K1); " if ordered else
                                                   "the GPUs share the unordered pairs (K1s); ") + what, "exchange": ex, "ordered": ordered}
        if k > 0 and not ex.endswith("one-gpu") and ndev < P:
            leg["skip"] = f"needs {P} GPUs, this box shows {ndev}"
        legs.append(leg)
    return legs


def failed_line(args, records, host):
    """No leg produced a measurement: still ONE JSON line, saying what was tried (value null, rc 1)."""
    return {"metric": "body-pair interactions/sec", "value": None, "unit": "pairs/s", "n_gpus": args.gpus, "steps": 0,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 pair math / f64 accumulate", "data": "synthetic",
            "config": workload_config(args.bodies, args.gpus, "no leg of the ladder completed"), "host": host,
            "error": "every leg of the ladder failed", "legs": records}


def finish_ladder_line(args, line, records, chosen, legs, extras, replicas):
    line["leg"] = legs[chosen]["name"]
    line["legs"] = records
    line["legs_note"] = ("the measured step ran as a ladder of bounded fresh child processes; `value`, `ms_per_step` and `roofline` "
                         "are those of leg `%s`, the first that completed" % line["leg"]
                         + ("" if not any(not r.get("ok") and "skipped" not in r for r in records[:len(records)]) else
                            "; legs before it failed or timed out and are recorded with their error"))
    if extras:
        line["variants"] = dict(extras, note="the other forms of the same step, each its own bounded child after the measured leg")
    if replicas is not None:
        line["replicas"] = replicas
    line.pop("stage", None)
    line["wall_s"]["process"] = round(time.perf_counter() - _T0, 2)
    line["wall_s"]["note"] = ("process = the orchestrating parent: device probe, every leg of the ladder (each a fresh child: imports, "
                              "set-up, warm-up, its timed region, its checks), variants, replicas; timed_region = the K steps "
                              "`value` comes from, inside the chosen leg")
    return line


def orchestrate_native(args):
    """`python3 bench.py --gpus P` typed as is (no launcher — the reference is one command on its GPUs too, hw5.cu:618).  THIS
    process never initialises HIP: it walks the ladder of native_legs(), each leg a fresh bounded child that drives the P GPUs
    through the C-ABI host, and prints the line of the first leg that completes with the failures before it recorded —
    a refused or wedged ncclReduceScatter costs its leg, not the line.  Then, untimed and bounded: the other legs as
    `variants`, the overlap toggle, and the reference's own multi-GPU mode (`replicas`)."""
    t0 = time.perf_counter()
    log = lambda msg: print(f"[bench] {msg}", file=sys.stderr, flush=True)  # noqa: E731
    if args.conservation or args.dump_rows:
        raise SystemExit("--conservation/--dump-rows: single rank or the torch host")
    ndev = args.gpus if args.leg_program else probe_device_count()  # (a test's stand-in legs bring their own "devices")
    if ndev <= 0:  # no usable GPU at all: fail loudly, with the library's own words, no ladder (there is no CPU path)
        rc, out, err = run_process([sys.executable, os.path.abspath(__file__)] + passthrough_args(args) +
                                   ["--host", "native", "--exchange", args.exchange or "rccl"], 300)
        sys.stderr.write(err)
        raise SystemExit(rc if rc else 1)
    legs = native_legs(args, ndev)
    program = args.leg_program.split() if args.leg_program else [sys.executable, os.path.abspath(__file__)]

    def runner(leg, timeout, steps=None, diagnostics=True):
        return run_process(program + passthrough_args(args, steps, diagnostics) + leg["argv"], timeout)

    records, chosen, line = run_ladder(legs, runner, args.legs_budget, args.leg_timeout, log)
    if line is None:
        emit(json.dumps(failed_line(args, records, "native")))
        raise SystemExit(1)
    extras, replicas = {}, None
    if not args.no_diagnostics:
        k_ab = min(5, max(2, args.steps))
        win = legs[chosen]
        todo = [lg for lg in legs[chosen + 1:] if not lg.get("skip")]
        if not args.overlap and args.bodies // args.gpus % 256 == 0 and not win["exchange"].startswith("host"):
            # the two-phase step that hides the all-gather (ordered pairs; the host-staged exchange has no overlapped form)
            todo.append({"name": "ordered_pairs_" + win["exchange"].replace("-", "_") + "_overlap",
                         "argv": ["--host", "native", "--exchange", win["exchange"], "--ordered-pairs", "--overlap"]})
        for lg in todo:
            if time.perf_counter() - t0 > args.legs_budget - 60:
                extras[lg["name"]] = {"skipped": "out of time budget"}
                continue
            t1 = time.perf_counter()
            rc, out, err = runner(lg, min(args.leg_timeout, 180), steps=k_ab, diagnostics=False)
            rec, ln = leg_record(lg["name"], rc, out, err, min(args.leg_timeout, 180), time.perf_counter() - t1)
            extras[lg["name"]] = summarise_leg_line(ln) if rec["ok"] else {k: rec[k] for k in ("error", "timeout", "stderr_tail") if k in rec}
        replicas = replicas_check([0, 0] if legs[chosen]["exchange"].endswith("one-gpu") else list(range(args.gpus)))
    emit(json.dumps(finish_ladder_line(args, line, records, chosen, legs, extras, replicas)))


def torch_legs(args, world, ndev):
    """The ladder under torch.distributed.run (one parent per rank, each leg = one fresh child per rank forming its own process
    group on a fresh port): the ranks share the unordered pairs (reduce_scatter_tensor + all_gather_into_tensor), then
    north_star's literal scheme (ordered pairs, all-gather only), then — no RCCL at all — the native C-ABI host with the
    copy-engine exchange, run by rank 0 alone while the other parents wait on the host."""
    req_ordered = bool(args.ordered_pairs or args.overlap or args.exchange == "ring")
    common = ["--backend", args.backend] + (["--single-device"] if args.single_device else []) + \
             (["--exchange", args.exchange] if args.exchange else []) + (["--overlap"] if args.overlap else [])
    coll = {"nccl": "RCCL", "gloo": "gloo (rehearsal)"}.get(args.backend, args.backend)
    legs = []
    for ordered in ([False, True] if not req_ordered else [True]):
        name = ("ordered_pairs" if ordered else "shared_pairs") + "_" + args.backend + \
               ("_ring" if args.exchange == "ring" else "") + ("_overlap" if args.overlap else "")
        legs.append({"name": name, "host": "torch", "argv": common + (["--ordered-pairs"] if ordered else []), "ordered": ordered,
                     "what": f"one process per GPU, torch.distributed: " + (f"ordered pairs (K1), {coll} all-gather of positions only "
                             "(north_star's scheme)" if ordered else f"the ranks share the unordered pairs (K1s), {coll} reduce-scatter of "
                             "partial forces + all-gather of positions")})
    if args.exchange != "ring":
        ex = "copy-one-gpu" if (args.single_device or ndev < world) else "copy"
        for ordered in ([False, True] if not req_ordered else [True]):
            legs.append({"name": "native_" + ("ordered_pairs" if ordered else "shared_pairs") + "_" + ex.replace("-", "_") +
                                 ("_overlap" if args.overlap else ""), "host": "native", "exchange": ex, "ordered": ordered,
                         "argv": ["--host", "native", "--exchange", ex] + (["--ordered-pairs"] if ordered else []) +
                                 (["--overlap"] if args.overlap else []),
                         "what": "the C-ABI host (ONE process for all GPUs, run by rank 0), copy-engine exchange: no RCCL involved"})
        if not args.overlap:
            hx = "host-one-gpu" if ex == "copy-one-gpu" else "host"
            legs.append({"name": "native_ordered_pairs_" + hx.replace("-", "_"), "host": "native", "exchange": hx, "ordered": True,
                         "argv": ["--host", "native", "--exchange", hx, "--ordered-pairs"],
                         "what": "the C-ABI host, all-gather through a pinned host array: per-device copies only, no peer-to-peer, no "
                                 "RCCL (last resort)"})
    return legs


def orchestrate_torch(args, world):
    """`python -m torch.distributed.run --nproc-per-node P bench.py --gpus P` (the driver's form).  Every launched process is a
    PARENT that never initialises HIP; the parents keep in step over a host-side gloo group (the launcher's store) and walk
    the ladder of torch_legs() together: for every leg each parent starts ONE fresh child — its rank of the leg, on a fresh
    rendezvous port — bounded by the leg's time limit, and stops it early when another rank's child has failed.  Rank 0 prints
    the line of the first leg that completed, with the legs before it recorded."""
    import datetime
    import socket

    import torch.distributed as dist
    t0 = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the torch host runs one rank per GPU under "
                         f"torch.distributed.run (or type the command without a launcher for the native host)")
    if args.exchange in NATIVE_EXCHANGES:
        raise SystemExit(f"--exchange {args.exchange} belongs to the native host (run without a launcher)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    try:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=max(1800.0, 4 * args.legs_budget)))
    except Exception as e:  # noqa: BLE001
        # the parents cannot even form their host-side group (the launcher's store is unreachable): the torch legs need it,
        # the native host does not — rank 0 walks the native ladder alone (one process for all GPUs), the others step aside
        print(f"[bench] rank {rank}: no host-side group among the parents ({type(e).__name__}: {e}); "
              + ("rank 0 runs the native host's ladder alone" if rank == 0 else "leaving the GPUs to rank 0"), file=sys.stderr, flush=True)
        if rank != 0:
            return
        args.exchange = None
        return orchestrate_native(args)
    log = (lambda msg: print(f"[bench] {msg}", file=sys.stderr, flush=True)) if rank == 0 else None

    def bcast(x):
        box = [x]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    try:
        store = dist.distributed_c10d._get_default_store()
        store.check(["nb_probe"])
    except Exception:  # noqa: BLE001  (no early stop then: every rank waits for its own time limit)
        store = None
    ndev = bcast((args.gpus if args.leg_program else probe_device_count()) if rank == 0 else None)
    legs = torch_legs(args, world, ndev)
    program = args.leg_program.split() if args.leg_program else [sys.executable, os.path.abspath(__file__)]
    serial = [0]

    def runner(leg, timeout, steps=None, diagnostics=True, parity=None):
        serial[0] += 1
        tag = f"nb_leg_failed_{serial[0]}"
        timeout = bcast(timeout)
        argv = program + passthrough_args(args, steps, diagnostics, parity) + leg["argv"]
        if leg["host"] == "native":  # rank 0 alone drives all GPUs; the other parents wait here, their GPUs idle
            return bcast(run_process(argv, timeout) if rank == 0 else None)
        port = bcast(free_port() if rank == 0 else None)
        env = child_env(RANK=str(rank), LOCAL_RANK=os.environ.get("LOCAL_RANK", str(rank)), WORLD_SIZE=str(world),
                        LOCAL_WORLD_SIZE=os.environ.get("LOCAL_WORLD_SIZE", str(world)),
                        MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port))

        def others_failed():
            try:
                return bool(store is not None and store.check([tag]))
            except Exception:  # noqa: BLE001
                return False

        rc, out, err = run_process(argv, timeout, env, others_failed)
        if rc != 0 and store is not None:
            try:
                store.set(tag, str(rank))
            except Exception:  # noqa: BLE001
                pass
        every = [None] * world
        dist.all_gather_object(every, (rc, " | ".join(ln for ln in err.strip().splitlines()[-4:] if ln.strip())[-400:]))
        if rank == 0:
            bad = [f"rank {r}: " + ("killed at the time limit / stopped" if c is None else f"rc={c}") + (f" ({tail})" if tail else "")
                   for r, (c, tail) in enumerate(every) if r > 0 and c != 0]
            if bad:
                err = err + "\n" + "\n".join(bad)
                if rc == 0 and last_json_line(out) is None:
                    rc = -1
        return bcast((rc, out, err) if rank == 0 else None)

    records, chosen, line = run_ladder(legs, runner, args.legs_budget, args.leg_timeout, log, sync=bcast)
    if line is None:
        if rank == 0:
            emit(json.dumps(failed_line(args, records, "torch")))
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(1)
    extras, replicas = {}, None
    if not args.no_diagnostics:
        k_ab = min(5, max(2, args.steps))
        todo = list(legs[chosen + 1:])
        if legs[chosen]["host"] == "torch" and args.exchange != "ring":
            # the same workload through the C-ABI host over RCCL (what `python3 bench.py --gpus P` measures), with its oracle check
            ex = "copy-one-gpu" if (args.single_device or ndev < world) else "rccl"
            todo.insert(0, {"name": "native_host", "host": "native", "parity": True, "steps": min(10, max(2, args.steps)),
                            "argv": ["--host", "native", "--exchange", ex]})
        for lg in todo:
            if bcast(time.perf_counter() - t0 > args.legs_budget - 60):
                extras[lg["name"]] = {"skipped": "out of time budget"}
                continue
            t1 = time.perf_counter()
            limit = min(args.leg_timeout, 180)
            rc, out, err = runner(lg, limit, steps=lg.get("steps", k_ab), diagnostics=False, parity=lg.get("parity", False))
            rec, ln = leg_record(lg["name"], rc, out, err, limit, time.perf_counter() - t1)
            if rec["ok"] and lg["name"] == "native_host":
                keep = ("value", "ms_per_step", "n_gpus", "steps", "host", "exchange", "ranks", "kernel_ms_per_rank",
                        "non_kernel_ms_per_step", "parity_spot")
                extras[lg["name"]] = dict({k: ln[k] for k in keep if k in ln}, roofline_frac=ln.get("roofline", {}).get("frac"),
                                          what="the same workload through the C-ABI host (nb_sharded_*: ONE process, ncclCommInitAll), "
                                               "run by rank 0 as a bounded child after the measured leg while the other ranks wait")
            else:
                extras[lg["name"]] = summarise_leg_line(ln) if rec["ok"] else {k: rec[k] for k in ("error", "timeout", "stderr_tail") if k in rec}
        if rank == 0:
            replicas = replicas_check([0, 0] if (args.single_device or ndev < 2) else list(range(world)))
    if rank == 0:
        native = extras.pop("native_host", None)
        line = finish_ladder_line(args, line, records, chosen, legs, extras, replicas)
        if native is not None:
            line["native_host"] = native
        emit(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()


def conservation(torch, sysm, n, sample=1024):
    """Integrals of motion of the state the system holds now (single rank): total momentum sum G m v exactly, kinetic
    energy exactly, potential energy from `sample` strided targets against ALL sources in fp64 (scaled by n/sample; the
    same sample before and after, so its change tracks the true change up to sampling noise).  Units: G*m (what the
    records carry), Plummer-softened potential as the force law (samples/nbody.cc:66-72)."""
    from nbody_amd import synthetic
    if sysm.acc64:
        pos, vel = sysm.pos64, sysm.vel64
    else:
        pos, vel = sysm.positions.double(), sysm.vel.double()
    gm = pos[:, 3]
    mom = (gm[:, None] * vel[:, :3]).sum(dim=0)
    scale = (gm[:, None] * vel[:, :3].abs()).sum(dim=0)
    kin = 0.5 * (gm * (vel[:, :3] ** 2).sum(dim=1)).sum()
    idx = torch.arange(sample, device=pos.device) * (n // sample)
    tq, tg = pos[idx, :3], gm[idx]
    phi = torch.zeros(sample, dtype=torch.float64, device=pos.device)
    chunk = 1 << 16
    eps2 = synthetic.EPS ** 2
    for j0 in range(0, n, chunk):
        sq, sg = pos[j0:j0 + chunk, :3], gm[j0:j0 + chunk]
        d2 = ((tq[:, None, :] - sq[None, :, :]) ** 2).sum(dim=2) + eps2
        w = sg[None, :] * torch.rsqrt(d2)
        w[(idx[:, None] == (torch.arange(j0, j0 + sq.shape[0], device=pos.device))[None, :])] = 0.0  # no self pair
        phi -= w.sum(dim=1)
    pot = 0.5 * (tg * phi).sum() * (n / sample)
    return {"momentum": [float(x) for x in mom], "momentum_scale": [float(x) for x in scale],
            "kinetic": float(kin), "potential_sampled": float(pot), "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bodies", type=int, default=1 << 20)
    ap.add_argument("--precision", choices=["f32", "f32acc64"], default="f32")
    ap.add_argument("--targets-per-lane", type=int, default=0)
    ap.add_argument("--j-split", type=int, default=0)
    ap.add_argument("--source-path", type=int, default=0, help="0 auto, 1 LDS tile, 2 SGPR/scalar loads (both: every ordered "
                    "pair), 3 every unordered pair once (K1s; auto picks it for a whole system of >= 28672 bodies on one GPU)")
    ap.add_argument("--wg-size", type=int, default=0, help="0 auto, 256, 512 (with --targets-per-lane 8), 1024 (4)")
    ap.add_argument("--overlap", action="store_true", help="multi-GPU: two-phase step, own-shard sources while the "
                    "all-gather of the other shards is in flight (SURVEY 8(f)-3); default off, see overlap_ab in the JSON")
    ap.add_argument("--exchange", choices=["in_place", "staged", "ring", "rccl", "copy", "copy-one-gpu", "host", "host-one-gpu"], default=None,
                    help="multi-GPU, torch host: in_place all-gather (default), all-gather from a cloned shard (staged), or "
                    "the ring pass (no rank holds all positions).  Native host: rccl (default, in-place ncclAllGather), copy "
                    "(peer copies on the copy engines) or copy-one-gpu (all ranks on device 0: rehearsal on a one-GPU box)")
    ap.add_argument("--host", choices=["auto", "native", "torch"], default="auto", help="auto: the torch host under "
                    "torch.distributed.run (WORLD_SIZE > 1) or with one GPU, the native C-ABI host (nb_sharded_*, one "
                    "process for all GPUs) when --gpus > 1 is typed without a launcher")
    ap.add_argument("--ordered-pairs", action="store_true", help="multi-GPU: every GPU evaluates every ordered pair of its own "
                    "targets (K1) even where the GPUs could share the unordered pairs of the system (K1s + reduce-scatter)")
    ap.add_argument("--no-diagnostics", action="store_true", help="multi-GPU: skip the untimed diagnostics after the timed "
                    "region (agreement check, overlap / exchange variants, native-host child, hw5 replicas)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-spot", action="store_true", help="skip the oracle spot check of the final state (N=1)")
    ap.add_argument("--no-live-pmc", action="store_true", help="N=1: do not re-run 2 steps three times under rocprofv3 --pmc "
                    "for this run's own HBM traffic / VALU-busy (the committed PMC profile is quoted instead)")
    ap.add_argument("--report-every", type=int, default=0, help="sustained runs (configs[4]): every R steps synchronise "
                    "and print steps done + running pairs/s to stderr")
    ap.add_argument("--time-box", type=float, default=0.0, help="sustained runs: stop at a report point once this many "
                    "seconds have elapsed; the JSON then carries the steps actually completed")
    ap.add_argument("--checkpoint", default="", help="sustained runs: NBODYST2 state file of the whole system, written by "
                    "rank 0 every --checkpoint-every steps (a collective; costs wall time inside the timed region)")
    ap.add_argument("--checkpoint-every", type=int, default=0)
    ap.add_argument("--resume", default="", help="start from this checkpoint instead of the synthetic initial state")
    ap.add_argument("--conservation", action="store_true", help="single rank: momentum, kinetic and sampled potential "
                    "energy before and after the run (their drift goes into the JSON)")
    ap.add_argument("--dump-rows", default="", help="single rank: save 64 strided rows of the final q, v (npz) — resumed "
                    "and uninterrupted runs are compared bit for bit with it")
    ap.add_argument("--lib", default="", help="experiments: load this build of libnbody_amd.so instead of the in-tree one "
                    "(same-device A/B of two kernel builds, e.g. make LIB=/tmp/x.so EXTRA=-DNB_K1_PAIR_GROUP=1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo + --single-device rehearses "
                    "the multi-rank path on a one-GPU box (RCCL refuses two ranks on one device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--deadline", type=float, default=120.0, help="multi-GPU: seconds one step (kernels + exchange) may take once "
                    "the host waits for it before the leg gives up with an error (native host: nb_sharded_set_deadline; torch "
                    "host: the collective timeout of the process group).  Raise it for systems whose step is longer")
    ap.add_argument("--leg-timeout", type=float, default=240.0, help="multi-GPU: wall-clock limit of one leg of the ladder (a "
                    "fresh child: imports, communicator, upload, warm-up, the timed steps, its checks)")
    ap.add_argument("--legs-budget", type=float, default=480.0, help="multi-GPU: wall-clock budget of the whole ladder plus the "
                    "untimed variants after it")
    ap.add_argument("--leg-child", action="store_true", help=argparse.SUPPRESS)    # this process IS a leg (started by the parent)
    ap.add_argument("--leg-program", default="", help=argparse.SUPPRESS)           # tests: run this instead of bench.py as a leg
    args = ap.parse_args()
    only_the_json_line_on_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and not args.leg_child:
        # the measured step of a multi-GPU line runs in bounded fresh children; this process orchestrates and never touches HIP
        if args.time_box or args.steps > 200:  # sustained runs: a leg may take as long as the run is asked to
            args.leg_timeout = max(args.leg_timeout, 2 * args.time_box + 600)
            args.legs_budget = max(args.legs_budget, 2 * args.leg_timeout)
        if world == 1 and args.host == "torch":
            raise SystemExit("--host torch runs one rank per GPU: launch with python -m torch.distributed.run --nproc-per-node P")
        return orchestrate_torch(args, world) if world > 1 else orchestrate_native(args)
    if world > 1 and args.host == "native":
        raise SystemExit("--host native is ONE process for all GPUs: run it without a launcher")
    if args.host == "native" or (args.host == "auto" and args.gpus > 1 and world == 1):
        return main_native(args)
    if args.exchange in NATIVE_EXCHANGES:
        raise SystemExit(f"--exchange {args.exchange} belongs to the native host (run without a launcher, or --host native)")
    args.exchange = args.exchange or "in_place"

    import torch
    import torch.distributed as dist

    import nbody_amd  # noqa: F401
    from nbody_amd import capi, synthetic
    from nbody_amd.distributed import ShardedSystem, hip_compute, shard_range, workspace_bytes
    if args.lib:
        capi.library_path = lambda: os.path.abspath(args.lib)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the torch host runs one rank per GPU under "
                         f"torch.distributed.run (or type the command without a launcher for the native host)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: nbody_amd has no CPU path")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        patience = datetime.timedelta(seconds=max(60.0, args.deadline))  # a wedged collective ends the leg instead of hanging it
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=patience)
        else:
            dist.init_process_group(args.backend, timeout=patience)
        # host-side control group: waiting at it parks no collective kernel on the GPUs (rank 0 runs child processes on them
        # during the diagnostics) and moves Python objects
        ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))

    n = args.bodies
    acc64 = args.precision == "f32acc64"
    lo, hi = shard_range(n, rank, world)
    first_step = 0
    if args.resume:
        hdr, pos, vel = ShardedSystem.load_checkpoint_shard(args.resume, rank, world)
        if hdr["n"] != n or hdr["dt"] != synthetic.DT or hdr["eps"] != synthetic.EPS:
            raise SystemExit(f"--resume: {args.resume} holds n={hdr['n']} dt={hdr['dt']} eps={hdr['eps']}")
        first_step = hdr["step"]
        if not acc64:
            pos, vel = pos.astype("float32"), vel.astype("float32")
    elif acc64:
        q, v, m = synthetic.bodies(n, lo, hi)
        import numpy as np
        pos = np.ascontiguousarray(np.concatenate([q.T, (synthetic.G * m)[:, None]], axis=1))
        vel = np.ascontiguousarray(np.concatenate([v.T, np.zeros((hi - lo, 1))], axis=1))
    else:
        pos, vel = synthetic.body4_f32(n, lo, hi)
    compute = hip_compute(acc64, args.targets_per_lane, args.j_split, args.source_path, args.wg_size)
    sysm = ShardedSystem(n, torch.from_numpy(pos), torch.from_numpy(vel), synthetic.EPS, synthetic.DT, device,
                         compute=compute, acc64=acc64, overlap=args.overlap, exchange=args.exchange,
                         shared_pairs=False if args.ordered_pairs else None)

    # --- kernel-only timing: HIP events on the stream the kernel is launched on (torch's current stream), recorded by
    #     ShardedSystem.step() around its launches
    kern_ms = sysm.kernel_events = []

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cons0 = conservation(torch, sysm, n) if (args.conservation and world == 1) else None
    for _ in range(args.warmup):
        sysm.step()
    barrier()
    kern_ms.clear()
    try:  # every rank samples its own GPU (world > 1: reported per rank in `ranks.per_rank[].power`, rank 0's also in `roofline.power`)
        pr_ = torch.cuda.get_device_properties(dev_index)
        power = PowerSampler(f"{getattr(pr_, 'pci_domain_id', 0):04x}:{getattr(pr_, 'pci_bus_id', 0):02x}:"
                             f"{getattr(pr_, 'pci_device_id', 0):02x}.0").start()
    except Exception:  # noqa: BLE001  (evidence, never a reason to fail the measurement)
        power = None
    t0 = time.perf_counter()
    steps_done = 0
    ckpt_s = []
    for k in range(args.steps):
        sysm.step()
        steps_done += 1
        if args.checkpoint and args.checkpoint_every and steps_done % args.checkpoint_every == 0:
            torch.cuda.synchronize()  # the step in flight is not part of the checkpoint's time
            tc = time.perf_counter()
            sysm.save_checkpoint(args.checkpoint, first_step + args.warmup + steps_done, synthetic.G)
            ckpt_s.append(time.perf_counter() - tc)
        if args.report_every and steps_done % args.report_every == 0 and steps_done < args.steps:
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            stop = torch.tensor([1.0 if (args.time_box and el > args.time_box) else 0.0])
            if world > 1:
                stop = stop.to(device if args.backend == "nccl" else "cpu")
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)  # every rank stops at the same step
            if rank == 0:
                print(f"[bench] {steps_done}/{args.steps} steps, {el:.1f} s, "
                      f"{n * (n - 1) * steps_done / el:.4e} pairs/s sustained", file=sys.stderr, flush=True)
            if float(stop.item()) > 0:
                break
    barrier()
    wall = time.perf_counter() - t0
    power = power.stop() if power is not None else None
    args.steps = steps_done
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    k_ms = sum(a.elapsed_time(b) for a, b in kern_ms) / len(kern_ms)
    assert torch.isfinite(sysm.positions).all(), "non-finite positions"
    cons1 = conservation(torch, sysm, n) if cons0 is not None else None
    if args.dump_rows and world == 1:
        import numpy as np
        idx = torch.arange(64, device=device) * (n // 64) + 17
        qd = (sysm.pos64 if acc64 else sysm.positions)[idx].cpu().numpy()
        vd = (sysm.vel64 if acc64 else sysm.vel)[idx].cpu().numpy()
        np.savez(args.dump_rows, idx=idx.cpu().numpy(), q=qd, v=vd, step=first_step + args.warmup + steps_done)

    # --- who ran: every rank's GPU, read by the rank itself, and every rank's own kernel time (host-side gloo group)
    kern_per_rank = rank_ids = None
    diag_errors = []
    if world > 1:
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            me = {"rank": rank, "device": dev_index, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
                  "pci_bus_id": f"{getattr(pr, 'pci_domain_id', 0):04x}:{getattr(pr, 'pci_bus_id', 0):02x}:"
                                f"{getattr(pr, 'pci_device_id', 0):02x}.0",
                  "compute_units": pr.multi_processor_count, "first_target": lo, "targets": hi - lo, "kernel_ms": k_ms}
            if power:
                me["power"] = {k: power[k] for k in ("mean_w", "max_w", "cap_w", "sclk_mhz_mean", "sclk_mhz_min", "samples")}
            ids = [None] * world
            dist.all_gather_object(ids, me, group=ctl)
            rank_ids, kern_per_rank = ids, [r["kernel_ms"] for r in ids]
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"ranks: {type(e).__name__}: {e}")

    # --- the line of the timed region (rank 0); a leg child prints it NOW, before any diagnostic can hang or fail
    out = None
    shared = bool(getattr(sysm, "shared_pairs", False))
    if rank == 0:
        pairs_step = n * (n - 1)
        # dominant kernel: a rank's force+kick-drift launch sequence = n_tgt x N pair evaluations, 20 flop each; with
        # several ranks the slowest one's time
        if kern_per_rank:
            k_ms = max(kern_per_rank)
        achieved = FLOP_PER_PAIR * sysm.n_tgt * (n - 1) / (k_ms * 1e-3) / 1e12
        n_cover = sysm.n_tgt if sysm.ring else n  # sources one launch sequence covers (ring pass: one travelling block)
        ws_bytes = workspace_bytes(n_cover, sysm.n_tgt, acc64, args.targets_per_lane, args.j_split, args.source_path,
                                   args.wg_size)  # what hip_compute allocates
        kname = capi.kernel_name_f32(n_cover, sysm.n_tgt, acc64, args.targets_per_lane, args.j_split, ws_bytes,
                                     source_path=args.source_path, wg_size=args.wg_size)
        tpl, jsp, wgs = capi.plan_f32(n_cover, sysm.n_tgt, acc64, args.targets_per_lane, args.j_split, ws_bytes,
                                      args.source_path, args.wg_size)
        if shared:  # the ranks share the unordered pairs: K1s; its launch shape from the library (nb_plan_shared_pairs_f32)
            _, jsp, _ = capi.plan_shared_pairs_f32(n, world, acc64)
            kname, tpl, wgs = f"nbody_force_sym_f32<{'true' if acc64 else 'false'}>", 8, 512
        coll = {"nccl": "RCCL (torch.distributed nccl)", "gloo": "gloo (host-staged: rehearsal, not RCCL)"}.get(args.backend, args.backend)
        out = {
            "metric": "body-pair interactions/sec",
            "value": pairs_step * args.steps / wall,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if not acc64 else "f32 pair math / f64 accumulate",
            "data": "synthetic",
            "config": workload_config(n, world, f"index-sharded x{world}, ring pass of float4 position blocks ({coll} send/recv)" if sysm.ring
                                      else f"index-sharded x{world}, "
                                      + (f"unordered pairs shared by the ranks, 1 {coll} reduce-scatter of partial forces + " if shared else "")
                                      + f"1 {coll} all-gather of float4 positions/step"),
            "host": "torch" if world > 1 else "single",
            "exchange": sysm.exchange_mode,
            "roofline": roofline_block(achieved, kname, k_ms, tpl, jsp, wgs, acc64, shared_pairs=shared),
        }
        if getattr(sysm, "shared_pairs_note", None):
            out["shared_pairs_note"] = sysm.shared_pairs_note
        if world > 1:
            out["host_detail"] = ("nbody_amd.distributed: one process per GPU, torch.distributed (backend "
                                  f"{args.backend}) for the collective only; kernels through the C ABI (nb_launch_step_f32)")
            out["pairs"] = ("every unordered pair once, shared by the ranks (K1s); every rank's partial force on all bodies is "
                            "reduce-scattered to the shard owners") if shared else "every ordered pair of a rank's own targets (K1)"
            out["overlap"] = bool(args.overlap)
        if args.single_device and world > 1:
            out["rehearsal"] = f"all {world} ranks on GPU 0 (they share the chip): the P > 1 host logic, not a {world}-GPU measurement"
        if rank_ids:
            out["ranks"] = {"count": world, "distinct_devices": len({(r["uuid"], r["pci_bus_id"]) for r in rank_ids}),
                            "backend": args.backend, "per_rank": rank_ids}
            out["kernel_ms_per_rank"] = kern_per_rank
            out["non_kernel_ms_per_step"] = wall / args.steps * 1e3 - k_ms
        if first_step:
            out["resumed_from_step"] = first_step
        if ckpt_s:
            out["checkpoints"] = {"count": len(ckpt_s), "seconds_each": [round(x, 2) for x in ckpt_s],
                                  "bytes": os.path.getsize(args.checkpoint), "inside_timed_region": True}
        if args.leg_child and world > 1:
            out["stage"] = "timed_region"
            emit(json.dumps(with_wall(out)))

    # --- untimed diagnostics (after the timed region; not part of `value`)
    exchange_ms = check = overlap_ab = spot = lds = sgpr = acc64_path = f64_path = None
    if world == 1 and not args.no_diagnostics and not sysm.ring and not args.lib:
        # the other kernels of the family on the same system, five untimed steps each, so that the adopted kernel's margin
        # comes from this run and not from a builder profile: the north star's named kernel — sources staged through an LDS
        # tile (source_path 1) — and K1 with sources broadcast from SGPRs (source_path 2), both evaluating every ORDERED pair
        def other_path(sp, what):
            main_compute, sysm.compute = sysm.compute, hip_compute(acc64, 0, 0, sp, 0)
            try:
                sysm.step()
                torch.cuda.synchronize()
                sysm.kernel_events = evs = []
                for _ in range(5):
                    sysm.step()
                torch.cuda.synchronize()
                ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            finally:
                sysm.compute, sysm.kernel_events = main_compute, None
            ws1 = workspace_bytes(n, n, acc64, 0, 0, sp, 0)
            return {"ms_per_step": ms, "frac": FLOP_PER_PAIR * n * (n - 1) / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                    "kernel": capi.kernel_name_f32(n, n, acc64, 0, 0, ws1, source_path=sp),
                    "plan": list(capi.plan_f32(n, n, acc64, 0, 0, ws1, sp, 0)), "steps": 5, "what": what}
        try:
            if args.source_path != 1:
                lds = other_path(1, "source_path=1: every ordered pair; 256-body float4 tiles through LDS, one coalesced 16-B load "
                                    "per lane per tile, broadcast ds_read_b128 in the pair loop; HIP events on the launch stream")
            if args.source_path != 2:
                sgpr = other_path(2, "source_path=2: every ordered pair (K1); sources broadcast from SGPRs (s_load_dwordx16), "
                                     "sliced launch + reducer; HIP events on the launch stream")
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"other kernels: {type(e).__name__}: {e}")
        # the other arithmetic modes of the same system (BASELINE configs[4] computes in fp32 with fp64 sums and fp64 q,v
        # masters; the testcases' mode is all-fp64), each with its own oracle check, so that their rates are the driver's too
        try:
            if not acc64 and args.source_path in (0, 3):
                acc64_path = other_precision_path(torch, n, device, "f32acc64")
            f64_path = other_precision_path(torch, min(n, 262144), device, "f64")
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"other precisions: {type(e).__name__}: {e}")
    if world > 1 and not args.no_diagnostics:
        # every rank takes the same path through these: an exception that all ranks raise alike (a refused argument) is
        # recorded in the JSON instead of losing the measurement above
        sysm.kernel_events = None
        if not sysm.ring and not args.no_parity_spot:
            # the published number carries its own proof: ONE more step, rows from every rank's shard against the oracle
            try:
                import numpy as np
                mine = [i for i in spot_rows(n, world) if lo <= i < hi]
                loc = torch.tensor([i - lo for i in mine], dtype=torch.long, device=device)
                pos0 = sysm.positions.clone() if rank == 0 else None
                vsrc = sysm.vel64 if acc64 else sysm.vel
                v0 = vsrc[loc, :3].double().cpu().numpy().T
                sysm.step()
                torch.cuda.synchronize()
                v1 = vsrc[loc, :3].double().cpu().numpy().T
                parts = [None] * world
                dist.all_gather_object(parts, (mine, v0, v1), group=ctl)
                if rank == 0:
                    idx = [i for part in parts for i in part[0]]
                    v0a, v1a = np.concatenate([p_[1] for p_ in parts], axis=1), np.concatenate([p_[2] for p_ in parts], axis=1)
                    p0 = pos0.cpu().numpy().astype(np.float64)
                    spot = step_rows_vs_oracle(np.ascontiguousarray(p0[:, :3].T), np.ascontiguousarray(p0[:, 3]), idx,
                                               v0a, v1a, acc64, world)
                del pos0
            except Exception as e:  # noqa: BLE001
                diag_errors.append(f"parity_spot: {type(e).__name__}: {e}")
        try:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            barrier()
            ev[0].record()
            for _ in range(20):
                sysm._exchange(sysm.positions)  # re-gathers the current positions: a no-op on the data
            ev[1].record()
            torch.cuda.synchronize()
            exchange_ms = ev[0].elapsed_time(ev[1]) / 20
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"exchange_ms: {type(e).__name__}: {e}")
        try:
            check = sharded_check(torch, dist, world, rank, device, dev_index, args.backend)
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"sharded_check: {type(e).__name__}: {e}")
        # the same system stepped in its other forms, a few steps each, wall clock (max over ranks): every rank its own
        # ORDERED pairs (K1) plain ("off") and with the two-phase step that hides the all-gather ("on"); and, where the timed
        # run used it, the ranks sharing the unordered pairs (K1s + reduce-scatter, "shared")
        try:
            ab = {}
            k_ab = min(5, max(2, args.steps))
            was_shared = sysm.shared_pairs
            for name, shared_, mode in (("shared", True, False), ("off", False, False), ("on", False, True)):
                if (mode and sysm.n_tgt % 256) or sysm.ring or (shared_ and not was_shared):
                    continue
                sysm._wait_gather()
                sysm.shared_pairs, sysm.overlap = shared_, mode
                sysm.step()
                barrier()
                t1 = time.perf_counter()
                for _ in range(k_ab):
                    sysm.step()
                barrier()
                w = torch.tensor([(time.perf_counter() - t1) / k_ab], dtype=torch.float64,
                                 device=device if args.backend == "nccl" else "cpu")
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                ab[name] = float(w.item()) * 1e3
            overlap_ab = ab
            sysm._wait_gather()
            sysm.shared_pairs = was_shared
        except Exception as e:  # noqa: BLE001
            diag_errors.append(f"overlap_ab: {type(e).__name__}: {e}")
        try:
            sysm._wait_gather()
        except Exception:  # noqa: BLE001
            pass
        sysm.overlap = args.overlap and world > 1

    if rank == 0:
        kname, jsp = out["roofline"]["kernel"], out["roofline"]["j_split"]
        traffic, reduce_share, valu_busy, pmc_tag = load_traffic(n, world, kname, jsp)
        traffic_source = (f"profiles/pmc_traffic.json ({pmc_tag}: builder's rocprofv3 PMC passes on this kernel, N and source "
                          f"split; not measured by this run)") if traffic else None
        live = None
        if world == 1 and not args.no_live_pmc and not args.lib:
            tail = ["--bodies", str(n), "--precision", args.precision, "--targets-per-lane", str(args.targets_per_lane),
                    "--j-split", str(args.j_split), "--source-path", str(args.source_path), "--wg-size", str(args.wg_size),
                    "--no-diagnostics"]
            live = live_pmc(tail, step_kernels(kname))
            if live and "error" not in live:
                traffic, valu_busy = live["hbm_bytes_per_step"], live["valu_busy"]
                traffic_source = ("measured by this run: three 2-step child runs under rocprofv3 --pmc (FETCH_SIZE x2 gfx950 "
                                  "correction + WRITE_SIZE; SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE); bytes = force kernel + "
                                  "reducer of one step (the launches kernel_ms spans), VALU-busy of the force kernel")
        r = out["roofline"]
        out["roofline"] = roofline_block(r["achieved"], kname, r["kernel_ms"], r["targets_per_lane"], jsp, r["wg_size"], acc64,
                                         traffic, traffic_source, live, valu_busy, reduce_share, shared_pairs=shared)
        if power:
            out["roofline"]["power"] = power
        if out["roofline"]["traffic_detail"]:
            out["roofline"]["traffic_detail"]["algorithmic"] = 56 * n  # 16N + 12N read, 12N + 16N written
        if world == 1 and kname.startswith("nbody_force_sym_f32"):
            # the workspace behind that traffic: what the timed launches used, and what the library takes by default at the
            # sizes of BASELINE configs[3] / [4] (round 5: one launch up to 2 GiB of slots, then batches within 720 B per body)
            out["roofline"]["workspace"] = {
                "bytes": workspace_bytes(n, n, acc64, args.targets_per_lane, args.j_split, args.source_path, args.wg_size),
                "bytes_per_body_beyond_2GiB": 720,
                "default_bytes_by_n": {str(k): capi.workspace_bytes_sym_f32(k, acc64) for k in (1 << 20, 1 << 22, 1 << 24)},
                "note": "pair slots of K1s (+ the running force of a batched step); rounds 1-4 took 26 GB at 2^22 and 52 GB at 2^24"}
        if lds is not None:
            out["lds_path"] = lds
        if sgpr is not None:
            out["ordered_pair_path"] = sgpr
        if acc64_path is not None:
            out["acc64_path"] = acc64_path
        if f64_path is not None:
            out["f64_path"] = f64_path
        if exchange_ms is not None:
            out["exchange_ms"] = exchange_ms  # one all-gather of float4[N] by itself, mean of 20
        if check is not None:
            out["sharded_check"] = check
        if spot is not None:
            out["parity_spot"] = spot
        if diag_errors:
            out["diagnostics_errors"] = diag_errors
        if overlap_ab:
            out["overlap_ab"] = {"ms_per_step": overlap_ab, "note": "same system, untimed diagnostic after the timed region: every "
                                 "rank its own ordered pairs (K1) without / with the two-phase step that hides the all-gather "
                                 "(off / on); shared = the ranks share the unordered pairs (K1s + reduce-scatter of forces)"}
        if cons0 is not None:
            e0, e1 = cons0["kinetic"] + cons0["potential_sampled"], cons1["kinetic"] + cons1["potential_sampled"]
            out["conservation"] = {
                "steps": args.warmup + args.steps, "before": cons0, "after": cons1,
                "momentum_drift_over_scale": max(abs(a - b) / sc for a, b, sc in
                                                 zip(cons0["momentum"], cons1["momentum"], cons0["momentum_scale"])),
                "energy_rel_change_sampled": (e1 - e0) / abs(e0),
                "note": "momentum and kinetic energy exact (fp64 sums over all bodies); potential from 1024 strided targets "
                        "x all sources in fp64, same sample before and after"}
        if world == 1 and not sysm.ring and not args.no_parity_spot:
            out["parity_spot"] = parity_spot(torch, sysm, n, acc64, dict(
                targets_per_lane=args.targets_per_lane, j_split=args.j_split, source_path=args.source_path,
                wg_size=args.wg_size))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(n)
        if world > 1:
            out["stage"] = "complete"
        emit(json.dumps(with_wall(out)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
