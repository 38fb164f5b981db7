"""bench.py's multi-GPU lines, and the native host on distinct GPUs.

`python3 bench.py --gpus P` typed without a launcher must run — the reference is one command on its GPUs (hw5.cu:618,
564-567) — through the C-ABI host (nb_sharded_*: one process, one in-place all-gather per GPU per step); under
torch.distributed.run the torch host runs and rank 0 adds the native host as a child.  Since round 5 the measured step of
both spellings runs as a LADDER of bounded fresh children (shared pairs over RCCL -> ordered pairs, all-gather only -> the
copy-engine exchange; tests/test_bench_ladder.py exercises the parent on the CPU): here the real legs run.  A one-GPU box
rehearses both with every rank on device 0 (`--exchange copy-one-gpu`, `--backend gloo --single-device`): exactly the code
paths of the driver's multi-GPU node except the collective library itself — and `--gpus 2 --exchange rccl` on one GPU is the
natural failure the ladder exists for.  The tests at the end need two GPUs and skip on the one-GPU box."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_LAUNCHER = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def _env():
    return {k: v for k, v in os.environ.items() if k not in _LAUNCHER}


def _line(p):
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]  # ONE JSON line and nothing else on stdout (librccl's
    return json.loads(lines[0])                                             # load banner goes to stderr)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_native_host_line_without_a_launcher(nb):
    """The command the verdict names: no launcher, two ranks, both on the one GPU."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "copy-one-gpu",
                        "--bodies", "131072", "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       timeout=900, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["n_gpus"] == 2 and r["host"] == "native" and r["steps"] == 3 and r["unit"] == "pairs/s"
    assert r["config"]["bodies"] == 131072 and r["scaling"] == "strong" and r["exchange"] == "copy-one-gpu"
    assert r["value"] == pytest.approx(131072 * 131071 * 3 / (r["ms_per_step"] * 3e-3), rel=1e-9)
    assert r["wall_s"]["process"] > r["wall_s"]["timed_region"] == pytest.approx(r["ms_per_step"] * 3e-3, abs=1e-3)
    # who ran: two ranks, one device here (they share it), each with its own shard
    rk = r["ranks"]
    assert rk["count"] == 2 and rk["distinct_devices"] == 1 and rk["exchange"] == "copy"
    assert [x["first_target"] for x in rk["per_rank"]] == [0, 65536] and all(x["targets"] == 65536 for x in rk["per_rank"])
    assert all(len(x["uuid"]) == 32 and x["pci_bus_id"] and x["name"] for x in rk["per_rank"])
    pw = r["roofline"].get("power_per_gpu")  # one entry per distinct GPU (amdgpu hwmon by PCI bus id); absent where sysfs hides it
    if pw:
        assert list(pw) == [rk["per_rank"][0]["pci_bus_id"]] and all(v is None or v["samples"] >= 1 for v in pw.values())
    # per-rank kernel time from HIP events on each rank's stream; the slowest rank prices the roofline
    assert len(r["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in r["kernel_ms_per_rank"])
    assert r["roofline"]["kernel_ms"] == max(r["kernel_ms_per_rank"]) and 0 < r["roofline"]["frac"] < 1
    # 131072 bodies in two shards of whole superblocks: the ranks share the unordered pairs (K1s + reduce-scatter of forces)
    assert r["roofline"]["kernel"] == "nbody_force_sym_f32<false>" and r["pairs"].startswith("every unordered pair once")
    # the line carries its own proof: rows of BOTH shards against the oracle
    ps = r["parity_spot"]
    assert ps["ok"] and ps["ranks_covered"] == 2 and ps["rows"] == 64 and ps["max_err_over_sum_abs"] < ps["tol"] == 1e-5
    assert r["sharded_check"]["ok"]
    # the measured step ran as a leg of the ladder: the request first, and it completed
    assert r["leg"] == "shared_pairs_copy_one_gpu" and [x["name"] for x in r["legs"]] == ["shared_pairs_copy_one_gpu"]
    assert r["legs"][0]["ok"] and "diagnostics_incomplete" not in r["legs"][0] and "stage" not in r
    assert "rehearsal" in r and "same-device copy" in r["config"]["parallelism"] and "ncclAllGather" not in r["config"]["parallelism"]
    # the other forms ran afterwards as bounded children: ordered pairs, and its overlapped step
    v = r["variants"]
    assert v["ordered_pairs_copy_one_gpu"]["kernel"].startswith("nbody_force_f32<") and v["ordered_pairs_copy_one_gpu"]["ms_per_step"] > 0
    assert v["ordered_pairs_copy_one_gpu_overlap"]["kernel"].startswith("nbody_force_f32<") and v["ordered_pairs_copy_one_gpu_overlap"]["overlap"]
    # the reference's own multi-GPU mode (two device slots, both this GPU): golden outputs
    assert r["replicas"]["b200"]["byte_identical"] and r["replicas"]["b1024"]["byte_identical"]


def test_a_refused_rccl_leg_costs_the_leg_not_the_line(nb):
    """`--gpus 2 --exchange rccl` on a box with ONE GPU.  RCCL cannot serve two ranks here (device 1 does
    not exist; one rank per GPU) — both RCCL legs fail by themselves — and the ladder ends on the copy-engine exchange with
    every rank on GPU 0: rc 0, ONE line, `legs[0].error` present, the value from the copy leg, its oracle spot check green."""
    if nb.capi.device_count() != 1:
        pytest.skip("a one-GPU box: the natural failure of an RCCL leg")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "rccl", "--bodies", "131072",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    r = _line(p)
    legs = r["legs"]
    assert legs[0]["name"] == "shared_pairs_rccl" and not legs[0]["ok"] and legs[0]["error"].startswith("rc=") and legs[0]["stderr_tail"]
    assert "no usable HIP device" in legs[0]["stderr_tail"] or "nccl" in legs[0]["stderr_tail"].lower()
    skipped = [x for x in legs if "skipped" in x]
    assert {x["name"] for x in skipped} == {"ordered_pairs_rccl", "shared_pairs_copy", "ordered_pairs_copy", "ordered_pairs_host"}  # need 2 GPUs
    assert r["leg"] == "shared_pairs_copy_one_gpu" and legs[-1]["name"] == r["leg"] and legs[-1]["ok"]
    assert r["exchange"] == "copy-one-gpu" and r["value"] == pytest.approx(131072 * 131071 * 3 / (r["ms_per_step"] * 3e-3), rel=1e-9)
    assert r["parity_spot"]["ok"] and r["parity_spot"]["ranks_covered"] == 2 and r["sharded_check"]["ok"]
    assert r["roofline"]["kernel"] == "nbody_force_sym_f32<false>" and "failed or timed out" in r["legs_note"]


def test_native_host_acc64_overlapped_line(nb):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--exchange", "copy-one-gpu",
                        "--bodies", "65536", "--steps", "2", "--warmup", "1", "--precision", "f32acc64", "--overlap",
                        "--no-diagnostics"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["n_gpus"] == 4 and r["overlap"] is True and r["ranks"]["count"] == 4
    ps = r["parity_spot"]
    assert ps["ok"] and ps["ranks_covered"] == 4 and ps["tol"] == 1e-6 and "variants" not in r and "replicas" not in r
    assert r["leg"] == "ordered_pairs_copy_one_gpu_overlap" and r["roofline"]["kernel"].startswith("nbody_force_f32<")


def test_torch_host_two_ranks_line(nb):
    """bench.py's world > 1 branch exactly as the driver launches it (torch.distributed.run, two ranks), rehearsed on the
    one GPU through gloo: every diagnostic present, none failed, the native host's line embedded."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--single-device", "--bodies", "32768", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=1200, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["n_gpus"] == 2 and r["host"] == "torch" and r["steps"] == 3 and "diagnostics_errors" not in r, r.get("diagnostics_errors")
    # 32768 bodies: too few for the shared pairs, so the first leg's ranks run K1 — and the line says gloo, not RCCL
    assert r["leg"] == "shared_pairs_gloo" and r["legs"][0]["ok"] and "diagnostics_incomplete" not in r["legs"][0]
    assert "gloo" in r["config"]["parallelism"] and "RCCL" not in r["config"]["parallelism"].replace("not RCCL", "") and "rehearsal" in r
    assert r["sharded_check"]["ok"] and r["exchange_ms"] > 0
    assert r["overlap_ab"]["ms_per_step"]["on"] > 0 and r["overlap_ab"]["ms_per_step"]["off"] > 0
    assert r["parity_spot"]["ok"] and r["parity_spot"]["ranks_covered"] == 2
    assert r["ranks"]["count"] == 2 and len(r["kernel_ms_per_rank"]) == 2
    assert r["roofline"]["kernel_ms"] == max(r["kernel_ms_per_rank"])
    nh = r["native_host"]
    assert "error" not in nh, nh
    assert nh["host"] == "native" and nh["n_gpus"] == 2 and nh["parity_spot"]["ok"] and nh["roofline_frac"] > 0
    assert r["replicas"]["b200"]["byte_identical"]


def test_single_gpu_line_reports_the_lds_kernel_and_step_traffic(nb):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "262144", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["n_gpus"] == 1 and r["host"] == "single" and r["parity_spot"]["ok"]
    # a whole system of >= 28672 bodies on one GPU: K1s (every unordered pair once) is what is timed; the two kernels that
    # evaluate every ordered pair are measured beside it by the same run
    assert r["roofline"]["kernel"] == "nbody_force_sym_f32<false>" and r["roofline"]["pair_evaluation"].startswith("each unordered")
    assert r["roofline"]["kernel_ms_spans"] == ["nbody_force_sym_f32<false>", "nbody_reduce_sym_f32<false, 0>"]
    k1 = r["ordered_pair_path"]
    assert k1["kernel"].startswith("nbody_force_f32<") and ", true, 512>" in k1["kernel"] and 0 < k1["frac"] < r["roofline"]["frac"]
    lds = r["lds_path"]
    assert lds["kernel"].startswith("nbody_force_f32<") and ", false, 256>" in lds["kernel"] and 0 < lds["frac"] < 1
    # the other arithmetic modes through the plain C-ABI context, each with its own oracle check (configs[4]'s fp32 pair math /
    # fp64 sums; the testcases' fp64 at large n against the fp64 vector peak)
    a64, f64 = r["acc64_path"], r["f64_path"]
    assert a64["kernel"] == "nbody_force_sym_f32<true>" and a64["parity"]["ok"] and a64["parity"]["tol"] == 1e-6 and 0 < a64["frac"] < 1
    assert f64["dtype"] == "f64" and f64["peak"] == 78.6 and f64["parity"]["ok"] and f64["parity"]["tol"] == 1e-12 and 0 < f64["frac"] < 1
    w = r["roofline"]["workspace"]
    assert w["bytes"] < 2.2e9 and w["default_bytes_by_n"]["4194304"] < 3.5e9 and w["default_bytes_by_n"]["16777216"] < 13e9
    assert r["roofline"]["flop_per_pair"] == 20 and r["roofline"]["flop_per_pair_executed"] == 13
    assert r["roofline"]["frac_executed"] == pytest.approx(r["roofline"]["frac"] * 13 / 20)
    pw = r["roofline"].get("power")  # amdgpu hwmon of this GPU, sampled during the timed region; absent where sysfs hides it
    if pw:
        assert pw["samples"] >= 1 and 50 < pw["mean_w"] <= pw["max_w"] < 2000 and (pw["sclk_mhz_min"] or 1) <= (pw["sclk_mhz_mean"] or 1) < 2600
    live = r["roofline"]["live_pmc"]
    if live and "error" not in live:  # rocprofv3 present: bytes of force kernel + reducer = the launches kernel_ms spans
        assert live["force"]["hbm_bytes"] > 0 and live["reducer"]["hbm_bytes"] > 0
        assert r["roofline"]["traffic"] == pytest.approx(live["force"]["hbm_bytes"] + live["reducer"]["hbm_bytes"])
        assert r["roofline"]["traffic_detail"]["algorithmic"] == 56 * 262144
        assert 0.5 < r["roofline"]["valu_busy"] <= 1.0 and 1.5 < r["roofline"]["clock_ghz_held"] < 2.6
        assert r["roofline"]["frac"] < r["roofline"]["frac_at_clock_held"] < 1.0  # the chip holds less than the nominal 2.4 GHz


# ---------------------------------------------------------------- two distinct GPUs (skipped on the one-GPU box)

def _two_gpus(nb):
    if nb.capi.device_count() < 2:
        pytest.skip("needs two GPUs: the cross-device halves of the exchanges (hipMemcpyPeerAsync, RCCL with 2 ranks)")


@pytest.mark.parametrize("n", [32768, 262144])  # ordered pairs (K1) / the GPUs share the unordered pairs (K1s + reduce-scatter)
@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("exchange", ["copy", "rccl", "host"])
def test_two_distinct_gpus_follow_nb_step(nb, oracle, exchange, overlap, n):
    """Devices [0, 1]: peer copies over xGMI / a two-rank RCCL all-gather (and, at 262144 bodies, reduce-scatter of the
    partial forces), plain and overlapped, against nb_step and oracle rows from both shards.  NOT EXECUTED on the builder's
    one-GPU boxes."""
    _two_gpus(nb)
    if exchange == "host" and overlap:
        pytest.skip("the host-staged exchange is not overlapped")
    from test_gpu_sharded_native import _oracle_one_step
    c, syn = nb.capi, nb.synthetic
    dt = 1e-2
    q, v, m = syn.bodies(n)
    with c.Context(n, c.NB_F32, 0, G=syn.G, eps=syn.EPS, dt=dt) as ctx:
        ctx.set_state(q, v, m)
        ctx.step(1, 3)
        q_ref, _ = ctx.get_state()
    with c.Sharded(n, [0, 1], c.NB_F32, G=syn.G, eps=syn.EPS, dt=dt, overlap=overlap, exchange=exchange) as sh:
        ids = [sh.rank_info(r) for r in range(2)]
        assert ids[0]["uuid"] != ids[1]["uuid"] and [i["comm_ranks"] for i in ids] == [2, 2]
        assert [i["comm_device"] for i in ids] == [0, 1]
        sh.set_state(q, v, m)
        sh.step(1)
        q1, v1 = sh.get_state()
        sh.step(2)
        q3, _ = sh.get_state()
    per = n // 2
    rows = [(r * per + off, 8) for r in range(2) for off in (0, per // 2 + 3, per - 8)]
    idx, qo, vo = _oracle_one_step(oracle, syn, q, v, m, dt, rows, f32_start=True)
    assert np.abs(v1[:, idx] - vo).max() < 2e-6 and np.abs(q1[:, idx] - qo).max() < 2e-7
    assert np.abs(q3 - q_ref).max() < 5e-7


def test_reference_mode_on_two_distinct_gpus(nb):
    """NB_DEVICES=0,1 bin/hw5: P2's arrival snapshot crosses a real device boundary (hw5.cu:482-484)."""
    _two_gpus(nb)
    sys.path.insert(0, ROOT)
    import bench
    r = bench.replicas_check([0, 1])
    assert r["b1024"]["byte_identical"] and r["b200"]["byte_identical"], r


def test_ladder_on_two_distinct_gpus(nb):
    """`python3 bench.py --gpus 2` on two real GPUs: the first leg — shared pairs, ncclReduceScatter + ncclAllGather over two
    ranks — should simply work; whatever it does, the line must come, prove itself against the oracle, and say which leg it
    is.  NOT EXECUTED on the builder's one-GPU boxes."""
    _two_gpus(nb)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "262144", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=1200, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["value"] > 0 and r["parity_spot"]["ok"] and r["ranks"]["distinct_devices"] == 2 and "rehearsal" not in r
    assert r["leg"] in [x["name"] for x in r["legs"] if x.get("ok")]
    if r["leg"] == "shared_pairs_rccl":
        assert r["ranks"]["comm_ranks_seen_by_every_rank"] == [2] and r["roofline"]["kernel"] == "nbody_force_sym_f32<false>"
    for name in ("ordered_pairs_rccl", "shared_pairs_copy", "ordered_pairs_copy", "ordered_pairs_host"):
        assert name in r["variants"] or name == r["leg"] or name in [x["name"] for x in r["legs"]]


def test_a_wedged_exchange_ends_its_leg_with_a_message(nb):
    """What a collective that never completes looks like from the outside, produced here by an allowance no step can meet
    (--deadline 1e-4 s for steps of milliseconds): every leg's child gives up by ITSELF — "exchange timed out" from the native
    host's bounded waits, rc != 0, within seconds, not at the leg's time limit — the ladder walks on, and with every leg
    failing the line still comes: value null, the legs recorded, rc 1."""
    import time
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "copy-one-gpu", "--bodies", "262144",
                        "--steps", "3", "--warmup", "1", "--deadline", "1e-4", "--leg-timeout", "120"], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert p.returncode == 1 and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["value"] is None and r["error"] == "every leg of the ladder failed"
    assert [x["name"] for x in r["legs"]] == ["shared_pairs_copy_one_gpu", "ordered_pairs_copy_one_gpu", "ordered_pairs_host_one_gpu"]
    for leg in r["legs"]:
        assert leg["error"].startswith("rc=") and "timed out after 0.0001 s" in leg["stderr_tail"] and leg["seconds"] < 60, leg
    assert time.perf_counter() - t0 < 150


def test_launcher_spelling_survives_refused_nccl_legs(nb):
    """The driver's form on a box with ONE GPU and the real backend: `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
    (nccl).  Rank 1's child has no device 1 and dies; rank 0's child would wait in the rendezvous for ever and is stopped early
    by its parent; both torch legs end that way.  The ladder goes on to the native host without RCCL — rank 0's parent alone,
    copy exchange, all ranks on GPU 0 — and rank 0 prints that line: rc 0."""
    import time
    if nb.capi.device_count() != 1:
        pytest.skip("a one-GPU box: the natural failure of the nccl legs")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--bodies", "131072", "--steps", "3", "--warmup", "1", "--no-diagnostics"],
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    r = _line(p)
    names = [x["name"] for x in r["legs"]]
    assert names == ["shared_pairs_nccl", "ordered_pairs_nccl", "native_shared_pairs_copy_one_gpu"], names
    for leg in r["legs"][:2]:
        assert not leg["ok"] and ("rank 1" in leg["stderr_tail"] or leg.get("error", "").startswith("rc=")), leg
        assert leg["seconds"] < 120, leg  # stopped early, not at the 240 s limit
    assert r["leg"] == "native_shared_pairs_copy_one_gpu" and r["host"] == "native" and r["parity_spot"]["ok"] and "rehearsal" in r
    assert time.perf_counter() - t0 < 400


def test_native_legs_checkpoint_and_resume(nb, tmp_path):
    """`bench.py --gpus P --checkpoint f --checkpoint-every K` / `--resume f` through the native legs (nb_sharded_save_state /
    nb_sharded_load_state): the checkpoints are written inside the timed region and said to be; a second run resumes from the
    last one and still proves itself against the oracle."""
    ck = str(tmp_path / "run.nbst")
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "copy-one-gpu", "--bodies", "65536",
              "--no-diagnostics"]
    p = subprocess.run(common + ["--steps", "4", "--warmup", "1", "--checkpoint", ck, "--checkpoint-every", "2"],
                       capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["checkpoints"]["count"] == 2 and r["checkpoints"]["inside_timed_region"] and r["checkpoints"]["bytes"] == 64 + 65536 * 57
    assert nb.capi.state_file_info(ck) == (65536, nb.capi.NB_F32, 5)  # warm-up + 4 steps
    p = subprocess.run(common + ["--steps", "2", "--warmup", "0", "--resume", ck], capture_output=True, text=True, timeout=600,
                       env=_env(), cwd=ROOT)
    r = _line(p)
    assert r["resumed_from_step"] == 5 and r["steps"] == 2 and r["parity_spot"]["ok"]
