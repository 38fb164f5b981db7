"""nb_sharded_*: the index-sharded stepper behind the C ABI (one process, P GPUs, in-place all-gather per GPU per
step).  A one-GPU box runs P = 1 through the RCCL code (communicator, all-gather call and all), where the trajectory
must equal nb_step's bit for bit, and P = 2, 4 through NB_SHARDED_COPY_EXCHANGE with every rank on device 0: the same
step_once — per-rank streams, stepped/gathered events, the FIRST/MIDDLE/LAST phase order of the overlapped step, the
ping-pong and the all-gather protocol — against nb_step AND against the oracle.  RCCL with P > 1 is the driver's
multi-GPU run (bin/nbody_bench N steps warmup f32 P)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["NB_F32", "NB_F32_ACC64"])
@pytest.mark.parametrize("n", [4096 + 3, 131072])
def test_one_device_equals_nb_step_bitwise(nb, precision, n):
    c, syn = nb.capi, nb.synthetic
    prec = getattr(c, precision)
    q, v, m = syn.bodies(n)
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=1e-2) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        q0, v0 = ctx.get_state()
    with c.Sharded(n, [0], prec, G=syn.G, eps=syn.EPS, dt=1e-2) as sh:
        sh.set_state(q, v, m)
        sh.step(2)
        ms = sh.step_timed(1)
        q1, v1 = sh.get_state()
        info = sh.info()
    assert np.array_equal(q0, q1) and np.array_equal(v0, v1)
    assert ms > 0 and info["devices"] == 1 and info["targets_per_device"] == n
    acc64 = prec == c.NB_F32_ACC64  # (from 28672 bodies on both pick K1s, which nb_create / nb_sharded_create size the workspace for)
    assert (info["targets_per_lane"], info["j_split"], info["wg_size"]) == \
        c.plan_f32(n, n, acc64, workspace_bytes=c.workspace_bytes_sym_f32(n, acc64) or c.workspace_bytes_f32(n, acc64) // 18 * 66)
    # (below K1s' threshold both hosts give K1 room for up to 64 slices in one launch: 66 records per body, not the minimum 18)
    assert np.abs(q1 - q).max() > 1e-6


def _oracle_one_step(oracle, syn, q, v, m, dt, rows, f32_start):
    """q,v of `rows` after ONE run_step from the oracle's fp64 accelerations (samples/nbody.cc:56-88)."""
    gm = (syn.G * m).astype(np.float32).astype(np.float64) / syn.G  # the kernels round G*m once to fp32
    q0 = q.astype(np.float32).astype(np.float64) if f32_start else q
    v0 = v.astype(np.float32).astype(np.float64) if f32_start else v
    # sources are always the fp32-rounded positions (also in NB_F32_ACC64: the pair loop reads the float4 copies)
    idx = np.concatenate([np.arange(lo, lo + cnt) for lo, cnt in rows])
    a = oracle.accel_rows_at(np.ascontiguousarray(q.astype(np.float32).astype(np.float64)), gm, syn.G, syn.EPS, idx)
    v1 = v0[:, idx] + a * dt
    return idx, q0[:, idx] + v1 * dt, v1


@pytest.mark.parametrize("precision", ["NB_F32", "NB_F32_ACC64"])
@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_sharing_one_gpu_follow_nb_step_and_the_oracle(nb, oracle, ranks, overlap, precision):
    """Every P > 1 line of the native host on one GPU (copy exchange, all ranks on device 0), N = 16384:
    3 steps vs the unsharded nb_step (same arithmetic, sums cut at other places: fp32 rounding), and the first step
    vs oracle rows taken from every rank's shard."""
    c, syn = nb.capi, nb.synthetic
    prec = getattr(c, precision)
    n, dt = 16384, 1e-2
    q, v, m = syn.bodies(n)
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=dt) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        q_ref, v_ref = ctx.get_state()
    with c.Sharded(n, [0] * ranks, prec, G=syn.G, eps=syn.EPS, dt=dt, overlap=overlap, exchange="copy") as sh:
        info = sh.info()
        assert info["devices"] == ranks and info["targets_per_device"] == n // ranks
        sh.set_state(q, v, m)
        sh.step(1)
        q1, v1 = sh.get_state()
        sh.step(2)
        q3, v3 = sh.get_state()
    per = n // ranks
    rows = [(r * per + off, 8) for r in range(ranks) for off in (0, per // 2 + 3, per - 8)]
    idx, qo, vo = _oracle_one_step(oracle, syn, q, v, m, dt, rows, f32_start=(precision == "NB_F32"))
    # |a| = O(1), dt = 1e-2: a force error of 1e-5 * sum|a_ij| would move v by 1e-7 * O(10); fp32 state rounds at 6e-8
    tol_v, tol_q = (2e-6, 2e-7) if precision == "NB_F32" else (2e-6, 3e-8)
    assert np.abs(v1[:, idx] - vo).max() < tol_v, np.abs(v1[:, idx] - vo).max()
    assert np.abs(q1[:, idx] - qo).max() < tol_q, np.abs(q1[:, idx] - qo).max()
    assert np.abs(q3 - q_ref).max() < (5e-7 if precision == "NB_F32" else 1e-7), np.abs(q3 - q_ref).max()
    assert np.abs(v3 - v_ref).max() < 1e-5, np.abs(v3 - v_ref).max()
    assert np.abs(q3 - q).max() > 1e-6  # it moved


@pytest.mark.parametrize("precision", ["NB_F32", "NB_F32_ACC64"])
@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_share_the_unordered_pairs(nb, oracle, ranks, precision):
    """From 28672 bodies on (whole 4096-body superblocks per shard, step not overlapped) the GPUs share the UNORDERED pairs
    of the system: K1s per rank on the superblocks of its shard, the partial forces on all bodies reduce-scattered to the
    shard owners (copy exchange here: peer copies + an ordered sum in the kick-drift kernel), then the all-gather.  One
    step against oracle rows of every shard, three against nb_step (K1s on one GPU) and against the ordered-pair form."""
    c, syn = nb.capi, nb.synthetic
    prec = getattr(c, precision)
    n, dt = 1 << 18, 1e-2
    q, v, m = syn.bodies(n)
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=dt) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        q_ref, v_ref = ctx.get_state()
    with c.Sharded(n, [0] * ranks, prec, G=syn.G, eps=syn.EPS, dt=dt, exchange="copy") as sh:
        assert sh.kernel_name() == f"nbody_force_sym_f32<{'true' if prec == c.NB_F32_ACC64 else 'false'}>"
        info = sh.info()
        assert (info["targets_per_lane"], info["wg_size"]) == (8, 512) and info["j_split"] == -(-256 // (64 // ranks))
        sh.set_state(q, v, m)
        sh.step(1)
        q1, v1 = sh.get_state()
        sh.step(2)
        q3, v3 = sh.get_state()
    with c.Sharded(n, [0] * ranks, prec, G=syn.G, eps=syn.EPS, dt=dt, exchange="copy", ordered_pairs=True) as sh:
        assert sh.kernel_name().startswith("nbody_force_f32<")
        sh.set_state(q, v, m)
        sh.step(3)
        q3o, v3o = sh.get_state()
    per = n // ranks
    rows = [(r * per + off, 4) for r in range(ranks) for off in (0, per // 2 + 3, per - 4)]
    idx, qo, vo = _oracle_one_step(oracle, syn, q, v, m, dt, rows, f32_start=(precision == "NB_F32"))
    tol_v, tol_q = (2e-6, 2e-7) if precision == "NB_F32" else (2e-6, 3e-8)
    assert np.abs(v1[:, idx] - vo).max() < tol_v, np.abs(v1[:, idx] - vo).max()
    assert np.abs(q1[:, idx] - qo).max() < tol_q, np.abs(q1[:, idx] - qo).max()
    for qq, vv in ((q_ref, v_ref), (q3o, v3o)):
        assert np.abs(q3 - qq).max() < (5e-7 if precision == "NB_F32" else 1e-7), np.abs(q3 - qq).max()
        assert np.abs(v3 - vv).max() < 1e-5, np.abs(v3 - vv).max()
    assert np.abs(q3 - q).max() > 1e-6


@pytest.mark.parametrize("overlap", [False, True])
def test_overlap_flag_changes_nothing_but_the_schedule(nb, overlap):
    """Two ranks on one GPU, whole step vs own-shard-first phases: the same sums cut at another place."""
    c, syn = nb.capi, nb.synthetic
    n = 2 * 8192
    q, v, m = syn.bodies(n)
    out = []
    for ov in (False, overlap):
        with c.Sharded(n, [0, 0], c.NB_F32_ACC64, G=syn.G, eps=syn.EPS, dt=1e-2, overlap=ov, exchange="copy") as sh:
            sh.set_state(q, v, m)
            sh.step(4)
            out.append(sh.get_state())
    if not overlap:  # the same schedule twice: bit for bit (stream/event protocol is deterministic in its results)
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    else:
        assert np.abs(out[0][0] - out[1][0]).max() < 1e-9 and np.abs(out[0][1] - out[1][1]).max() < 1e-7


def test_ragged_shards(nb, oracle):
    """n/P not a multiple of the 256-body tile: the plain step takes it (tail lanes, ragged last source tile), the
    overlapped step — which cuts the sources at shard boundaries — refuses it at creation."""
    c, syn = nb.capi, nb.synthetic
    per = 4096 + 37
    n = 2 * per
    q, v, m = syn.bodies(n)
    with c.Sharded(n, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy") as sh:
        sh.set_state(q, v, m)
        sh.step(1)
        q1, v1 = sh.get_state()
    idx, qo, vo = _oracle_one_step(oracle, syn, q, v, m, 1e-2, [(0, 8), (per - 8, 16), (n - 8, 8)], f32_start=True)
    assert np.abs(v1[:, idx] - vo).max() < 2e-6 and np.abs(q1[:, idx] - qo).max() < 2e-7
    with pytest.raises(c.NBodyError) as e:
        c.Sharded(n, [0, 0], c.NB_F32, eps=syn.EPS, overlap=True, exchange="copy")
    assert e.value.code == c.NB_ERR_INVALID and "multiple of 256" in str(e.value)
    with pytest.raises(c.NBodyError) as e:
        c.Sharded(n + 1, [0, 0], c.NB_F32, eps=syn.EPS, exchange="copy")
    assert e.value.code == c.NB_ERR_INVALID and "divisible" in str(e.value)


def test_refusals(nb):
    c = nb.capi
    for kw, text in ((dict(devices=[0, 0]), "listed twice"), (dict(devices=[0], precision=c.NB_F64), "precision"),
                     (dict(devices=[0], eps=0.0), "eps")):
        args = dict(n=1024, devices=[0], precision=c.NB_F32)
        args.update(kw)
        with pytest.raises(c.NBodyError) as e:
            c.Sharded(**args)
        assert e.value.code == c.NB_ERR_INVALID and text in str(e.value)
    with pytest.raises(c.NBodyError) as e:
        c.Sharded(1024, [99])
    assert e.value.code == c.NB_ERR_NO_DEVICE
    with c.Sharded(1024, [0]) as sh, pytest.raises(c.NBodyError) as e:
        sh.step(1)  # no state yet
    assert e.value.code == c.NB_ERR_STATE


def test_compiled_host_runs_the_sharded_path(nb):
    """bin/nbody_bench N steps warmup precision gpus: the C++ host of the sharded API (no Python, no torch)."""
    p = subprocess.run([os.path.join(ROOT, "bin", "nbody_bench"), "32768", "3", "1", "f32", "1"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])  # (libdrm may print a notice first)
    assert r["gpus"] == 1 and r["n"] == 32768 and r["pairs_per_s"] > 1e11 and r["targets_per_gpu"] == 32768
    # four ranks on the one GPU, overlapped, copy exchange
    p = subprocess.run([os.path.join(ROOT, "bin", "nbody_bench"), "65536", "3", "1", "f32", "4", "1", "copy-one-gpu"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["gpus"] == 4 and r["exchange"] == "copy-one-gpu" and r["overlap"] == 1 and r["targets_per_gpu"] == 16384
    assert r["pairs_per_s"] > 1e11


@pytest.mark.parametrize("args", [["16384", "3", "1", "f32", "4", "1", "copy-one-gpu"], ["16384", "2", "1", "f32acc64", "2", "0", "copy-one-gpu"],
                                  ["16384", "2", "0", "f32", "1", "0", "rccl"],
                                  ["131072", "2", "1", "f32acc64", "4", "0", "copy-one-gpu"],   # the ranks share the unordered pairs
                                  ["131072", "2", "0", "f32", "1", "0", "rccl"],               # K1s on one GPU through RCCL's rank-1 path
                                  ["16384", "40", "1", "f32", "2", "0", "copy-one-gpu", "shared", "60"],   # bounded waits: the ring of 16 steps lapped
                                  ["16384", "3", "1", "f32acc64", "4", "0", "host-one-gpu", "ordered", "60"]])  # host-staged exchange
def test_sharded_host_under_host_asan(nb, args):
    """bin/asan/nbody_bench (`make asan`): the multi-GPU host — per-rank streams and events, phased launches, the copy
    exchange, the RCCL path with one rank — compiled with AddressSanitizer + UBSan (device code: the plain gfx950 build), on the GPU."""
    exe = os.path.join(ROOT, "bin", "asan", "nbody_bench")
    if not os.path.exists(exe):
        pytest.skip("bin/asan/nbody_bench not built (make asan)")
    p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["gpus"] == int(args[4]) and r["pairs_per_s"] > 1e9
    assert r["kernel"].startswith("nbody_force_sym_f32" if args[0] == "131072" else "nbody_force_f32<")


# ---------------------------------------------------------------- bounded waits (nb_sharded_set_deadline)

def test_a_generous_deadline_changes_nothing(nb):
    """With a deadline the host never blocks inside the runtime: it polls, and keeps at most 16 steps in flight.  40 steps
    (more than one lap of that ring) of 2 ranks, plain and overlapped, must equal the run without a deadline bit for bit."""
    c, syn = nb.capi, nb.synthetic
    n = 16384
    q, v, m = syn.bodies(n)
    for overlap in (False, True):
        out = []
        for deadline in (0.0, 120.0):
            with c.Sharded(n, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, overlap=overlap, exchange="copy",
                           deadline=deadline) as sh:
                sh.set_state(q, v, m)
                sh.step(23)
                _, kms = sh.step_profiled(17)
                assert all(k > 0 for k in kms)
                out.append(sh.get_state())
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_an_expired_deadline_is_an_error_not_a_hang(nb):
    """The allowance is per awaited step: 0.1 ms for steps of ~12 ms each can only expire.  The call returns NB_ERR_HIP with
    "timed out", the system refuses every later call (NB_ERR_STATE), and destroying it returns (it abandons what the GPU may
    still be using instead of blocking in hipFree).  This is what ends a bench leg whose collective never completes."""
    import time
    c, syn = nb.capi, nb.synthetic
    n = 1 << 18
    q, v, m = syn.bodies(n)
    sh = c.Sharded(n, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy")
    sh.set_state(q, v, m)
    sh.step(1)  # (first-launch costs out of the way)
    sh.set_deadline(1e-4)
    t0 = time.perf_counter()
    with pytest.raises(c.NBodyError, match="timed out after 0.0001 s") as e:
        sh.step(20)
    assert e.value.code == c.NB_ERR_HIP and time.perf_counter() - t0 < 5.0
    with pytest.raises(c.NBodyError) as e2:
        sh.step(1)
    assert e2.value.code == c.NB_ERR_STATE
    with pytest.raises(c.NBodyError):
        sh.get_state()
    t0 = time.perf_counter()
    sh.close()
    assert time.perf_counter() - t0 < 5.0
    # the GPU drains the abandoned steps by itself; a fresh system works
    with c.Sharded(16384, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy", deadline=60.0) as ok:
        qq, vv, mm = syn.bodies(16384)
        ok.set_state(qq, vv, mm)
        ok.step(2)
        assert np.isfinite(ok.get_state()[0]).all()


def test_the_unordered_pair_step_is_a_preference_not_a_requirement(nb):
    """nb_sharded_create on a GPU that cannot spare K1s' memory (here: most of the HBM is taken first) falls back to ordered
    pairs for every rank — create succeeds, `note` says why — instead of failing."""
    import torch
    c, syn = nb.capi, nb.synthetic
    n = 1 << 21   # one GPU: 1.85 GB of pair slots (in batches; 6.9 GB in one launch); two ranks: 1.9 GB + 2 x 32 MB each
    torch.cuda.empty_cache()  # (blocks cached by earlier tests would serve the big allocation below without taking device memory)
    free, _ = torch.cuda.mem_get_info(0)
    hog = torch.empty(max(0, free - (2 << 30)), dtype=torch.uint8, device="cuda:0")  # leave 2 GB: 3/4 of it is less than either
    try:
        with c.Sharded(n, [0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2) as sh:
            # one GPU holding the whole system keeps K1s: batches of superblocks that fit 3/4 of what is free (memory for speed)
            assert sh.kernel_name() == "nbody_force_sym_f32<false>" and "batches of superblocks" in sh.note
        with c.Sharded(n, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy") as sh:
            assert sh.kernel_name().startswith("nbody_force_f32<") and "3/4" in sh.note and "ordered pairs (K1) instead" in sh.note
        hog2 = torch.empty(max(0, torch.cuda.mem_get_info(0)[0] - (1 << 30)), dtype=torch.uint8, device="cuda:0")  # leave 1 GB: K1's
        # 0.7 GB of state and slices fit, not even batches of 16 superblocks (1.2 GB) do — (copy exchange: RCCL wants memory too)
        with c.Sharded(n, [0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy") as sh:
            assert sh.kernel_name().startswith("nbody_force_f32<") and "ordered pairs (K1) instead" in sh.note
        del hog2
    finally:
        del hog
        torch.cuda.empty_cache()
    with c.Sharded(n, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy") as sh:
        assert sh.kernel_name() == "nbody_force_sym_f32<false>" and sh.note == ""


# ---------------------------------------------------------------- the host-staged exchange (last resort of the ladder)

@pytest.mark.parametrize("precision", ["NB_F32", "NB_F32_ACC64"])
@pytest.mark.parametrize("ranks", [2, 4])
def test_host_staged_exchange_follows_nb_step_and_the_oracle(nb, oracle, ranks, precision):
    """NB_SHARDED_HOST_EXCHANGE: the all-gather through one pinned host array — D2H of every rank's own slot, a bounded host
    wait, H2D of the others' — only per-device copies, for a node whose peer-to-peer path is broken.  Ordered pairs (K1).  Same
    checks as the copy exchange: 3 steps against nb_step, the first against oracle rows of every shard; a system large enough
    for shared pairs still runs K1 here; the overlapped step is refused."""
    c, syn = nb.capi, nb.synthetic
    prec = getattr(c, precision)
    n, dt = 16384, 1e-2
    q, v, m = syn.bodies(n)
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=dt) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        q_ref, v_ref = ctx.get_state()
    with c.Sharded(n, [0] * ranks, prec, G=syn.G, eps=syn.EPS, dt=dt, exchange="host", deadline=60.0) as sh:
        assert sh.rank_info(ranks - 1)["exchange"] == "host" and sh.kernel_name().startswith("nbody_force_f32<")
        sh.set_state(q, v, m)
        sh.step(1)
        q1, v1 = sh.get_state()
        _, kms = sh.step_profiled(2)
        q3, v3 = sh.get_state()
    assert all(k > 0 for k in kms)
    per = n // ranks
    rows = [(r * per + off, 8) for r in range(ranks) for off in (0, per // 2 + 3, per - 8)]
    idx, qo, vo = _oracle_one_step(oracle, syn, q, v, m, dt, rows, f32_start=(precision == "NB_F32"))
    tol_v, tol_q = (2e-6, 2e-7) if precision == "NB_F32" else (2e-6, 3e-8)
    assert np.abs(v1[:, idx] - vo).max() < tol_v and np.abs(q1[:, idx] - qo).max() < tol_q
    assert np.abs(q3 - q_ref).max() < (5e-7 if precision == "NB_F32" else 1e-7) and np.abs(v3 - v_ref).max() < 1e-5
    assert np.abs(q3 - q).max() > 1e-6
    with c.Sharded(1 << 17, [0, 0], c.NB_F32, G=syn.G, eps=syn.EPS, dt=dt, exchange="host") as big:
        assert big.kernel_name().startswith("nbody_force_f32<")  # no shared pairs: a host-side sum of partial forces is not worth it
    with pytest.raises(c.NBodyError) as e:
        c.Sharded(n, [0, 0], prec, G=syn.G, eps=syn.EPS, dt=dt, exchange="host", overlap=True)
    assert e.value.code == c.NB_ERR_INVALID


# ---------------------------------------------------------------- checkpoints of the native host

@pytest.mark.parametrize("precision,n", [("NB_F32", 16384), ("NB_F32_ACC64", 16384), ("NB_F32", 1 << 17)])  # (2^17: the ranks share the pairs)
def test_sharded_checkpoint_resume_is_bitwise(nb, tmp_path, precision, n):
    """nb_sharded_save_state / nb_sharded_load_state: ONE NBODYST2 file for the whole system (configs[4] runs for
    hours).  Four steps in one go against two steps, a checkpoint, a NEW system with another rank count resumed from it, two more
    steps: bit for bit.  The file is a plain state file (nb_read_state_file reads it); one written under other parameters is refused."""
    c, syn = nb.capi, nb.synthetic
    prec = getattr(c, precision)
    q, v, m = syn.bodies(n)
    kw = dict(G=syn.G, eps=syn.EPS, dt=1e-2, exchange="copy")
    with c.Sharded(n, [0, 0], prec, **kw) as sh:
        sh.set_state(q, v, m)
        sh.step(4)
        q4, v4 = sh.get_state()
    path = str(tmp_path / "ck.nbst")
    with c.Sharded(n, [0, 0], prec, **kw) as sh:
        sh.set_state(q, v, m)
        sh.step(2)
        sh.save_state(path, step=2)
    hdr, qf, vf, mf, _ = c.read_state_file(path)
    assert (hdr["n"], hdr["step"], hdr["precision"], hdr["dt"]) == (n, 2, prec, 1e-2) and np.array_equal(mf, m)
    with c.Sharded(n, [0, 0], prec, **kw) as sh:   # the same rank count: every sum is cut at the same places
        assert sh.load_state(path) == 2
        q2, v2 = sh.get_state()
        assert np.array_equal(q2, qf) and np.array_equal(v2, vf)
        sh.step(2)
        qr, vr = sh.get_state()
    assert np.array_equal(qr, q4) and np.array_equal(vr, v4)
    with c.Sharded(n, [0, 0], prec, G=syn.G, eps=syn.EPS, dt=2e-2, exchange="copy") as other:
        with pytest.raises(c.NBodyError, match="does not match") as e:
            other.load_state(path)
        assert e.value.code == c.NB_ERR_INVALID
    with c.Context(n, prec, 0, G=syn.G, eps=syn.EPS, dt=1e-2) as ctx:   # ... and an unsharded context resumes from it too
        assert ctx.load_state(path) == 2
