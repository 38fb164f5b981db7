"""The ladder of `bench.py --gpus P` on the CPU: the measured step of a multi-GPU line runs as bounded
fresh child processes — shared pairs over RCCL, then ordered pairs with the all-gather only, then the copy-engine exchange — and
a leg that fails or hangs costs its leg, not the line.  Here the legs are played by tests/leg_stub.py (no GPU, no product code):
what is tested is the orchestrating parent — order, time limits, early stop of a leg's other ranks, what the line records.  The
real legs run in tests/test_gpu_bench_hosts.py."""
import json
import os
import socket
import subprocess
import sys
import time

from conftest import ROOT

STUB = f"{sys.executable} {os.path.join(ROOT, 'tests', 'leg_stub.py')}"
_LAUNCHER = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def _env():
    return {k: v for k, v in os.environ.items() if k not in _LAUNCHER and not k.startswith("TORCHELASTIC_")}


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_ladder_takes_the_first_leg_that_yields_a_line():
    bench = _bench()
    good = json.dumps({"value": 2.0, "stage": "complete"})
    seen = []

    def runner(leg, timeout):
        seen.append((leg["name"], timeout))
        return {"a": (3, "", "boom\nncclReduceScatter: unhandled system error"), "b": (None, "", ""),
                "c": (0, "noise\n" + good + "\n", ""), "d": (0, good, "")}[leg["name"]]

    legs = [{"name": "a"}, {"name": "skipme", "skip": "needs 8 GPUs"}, {"name": "b"}, {"name": "c"}, {"name": "d"}]
    records, chosen, line = bench.run_ladder(legs, runner, budget_s=100, leg_timeout_s=7)
    assert chosen == 3 and line == {"value": 2.0, "stage": "complete"} and [s[0] for s in seen] == ["a", "b", "c"]
    assert all(t <= 7 for _, t in seen)
    a, sk, b, c = records
    assert a["error"] == "rc=3" and "ncclReduceScatter" in a["stderr_tail"] and not a["ok"]
    assert sk == {"name": "skipme", "skipped": "needs 8 GPUs", "ok": False}
    assert b["timeout"].startswith("killed after 7") and not b["ok"] and c["ok"] and "diagnostics_incomplete" not in c
    # every leg fails: no line, everything recorded; a runner that cannot even start is a failed leg, not an exception
    def broken(leg, timeout):
        raise OSError("no such program")
    records, chosen, line = bench.run_ladder(legs[:1], broken, 100, 7)
    assert chosen is None and line is None and "OSError" in records[0]["stderr_tail"]
    # the limit of a leg leaves 90 s for every leg behind it (two hanging RCCL legs must not starve the copy legs) but is never
    # under a minute: four legs, 300 s, a runner that fails at once
    seen.clear()
    bench.run_ladder([{"name": "a"}, {"name": "a"}, {"name": "a"}, {"name": "a"}], runner, budget_s=300, leg_timeout_s=240)
    assert [round(t / 10) * 10 for _, t in seen] == [60, 120, 210, 240], seen
    # out of budget: legs are skipped, not started
    records, chosen, _ = bench.run_ladder([{"name": "a"}], runner, budget_s=0.0, leg_timeout_s=7)
    assert chosen is None and records[0]["skipped"] == "out of time budget"


def test_a_leg_killed_in_its_diagnostics_keeps_its_measurement():
    """A leg prints the line of its timed region at once; what hangs afterwards is cut off at the time limit and the line
    survives, flagged."""
    bench = _bench()
    prog = ("import sys, time, json; print(json.dumps({'value': 5.0, 'stage': 'timed_region'}), flush=True); "
            "print('checking...', file=sys.stderr, flush=True); time.sleep(600)")
    t0 = time.perf_counter()
    rc, out, err = bench.run_process([sys.executable, "-c", prog], timeout=2.0, env=_env())
    assert rc is None and time.perf_counter() - t0 < 30 and "checking" in err
    rec, line = bench.leg_record("x", rc, out, err, 2.0, 2.0)
    assert rec["ok"] and rec["diagnostics_incomplete"] and line["value"] == 5.0 and rec["timeout"].startswith("killed after 2")
    # the early stop used between the ranks of one leg
    t0 = time.perf_counter()
    rc, _, err = bench.run_process([sys.executable, "-c", "import time; time.sleep(600)"], timeout=60.0, env=_env(),
                                   should_abort=lambda: time.perf_counter() - t0 > 1.0)
    assert rc is None and time.perf_counter() - t0 < 20 and "another rank of this leg failed" in err


def _line(p):
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    return json.loads(lines[0])


def test_native_spelling_walks_the_ladder_to_the_copy_exchange():
    """`python3 bench.py --gpus 4` with legs that behave like a node whose RCCL is broken: shared pairs over RCCL dies, ordered
    pairs over RCCL hangs (killed at --leg-timeout), the copy-engine leg delivers — rc 0, ONE line, the failures in `legs`."""
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--leg-timeout", "4",
                        "--legs-budget", "120", "--leg-program", STUB], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    r = _line(p)
    assert p.returncode == 0 and time.perf_counter() - t0 < 120
    assert r["leg"] == "shared_pairs_copy" and r["value"] == 1.0e12 + 7 and r["n_gpus"] == 4 and "stage" not in r
    names = [x["name"] for x in r["legs"]]
    assert names == ["shared_pairs_rccl", "ordered_pairs_rccl", "shared_pairs_copy"]
    a, b, c = r["legs"]
    assert a["error"] == "rc=3" and "ncclReduceScatter" in a["stderr_tail"]
    assert b["timeout"].startswith("killed after 4") and c["ok"] and c["diagnostics_incomplete"]  # (the stub hangs after its line)
    # the forms that were not needed for the measurement ran afterwards, bounded, as variants
    assert r["variants"]["ordered_pairs_copy"]["value"] == 1.0e12 + 5 and "ordered_pairs_copy_overlap" in r["variants"]
    assert r["variants"]["ordered_pairs_host"]["value"] == 1.0e12 + 5
    assert "replicas" in r and r["wall_s"]["process"] >= r["wall_s"]["timed_region"]
    # asking for a form starts the ladder there; with every leg failing the line still comes, value null, rc 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange", "copy", "--ordered-pairs",
                        "--no-diagnostics", "--leg-program", STUB], capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    r = _line(p)
    assert p.returncode == 0 and r["leg"] == "ordered_pairs_copy" and [x["name"] for x in r["legs"]] == ["ordered_pairs_copy"]
    assert "variants" not in r and "replicas" not in r
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--leg-timeout", "2", "--legs-budget", "30",
                        "--leg-program", f"{sys.executable} -c raise(SystemExit(9))"], capture_output=True, text=True, timeout=300,
                       env=_env(), cwd=ROOT)
    r = _line(p)
    assert p.returncode == 1 and r["value"] is None and all(x["error"] == "rc=9" for x in r["legs"])
    assert [x["name"] for x in r["legs"]] == ["shared_pairs_rccl", "ordered_pairs_rccl", "shared_pairs_copy", "ordered_pairs_copy",
                                              "ordered_pairs_host"]  # the last resort: no peer-to-peer, no RCCL


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_launcher_spelling_parents_walk_the_ladder_together():
    """The driver's form, two ranks on the CPU: every launched process is a parent; the parents agree over gloo, give each leg
    a fresh rendezvous port and one child per rank, and stop a leg's surviving rank when another has died (rank 1 of the
    shared-pairs leg exits at once, rank 0 would sleep for an hour).  The ordered-pairs leg completes: its line, rc 0."""
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
                        "6", "--backend", "gloo", "--single-device", "--leg-timeout", "12", "--legs-budget", "200",
                        "--leg-program", STUB], capture_output=True, text=True, timeout=400, env=_env(), cwd=ROOT)
    r = _line(p)
    assert p.returncode == 0, p.stderr[-2000:]
    assert time.perf_counter() - t0 < 100
    assert r["leg"] == "ordered_pairs_gloo" and r["value"] == 1.0e12 + 6 and r["stub_world"] == 2
    a, b = r["legs"]
    assert a["name"] == "shared_pairs_gloo" and not a["ok"] and "rank 1: rc=3" in a["stderr_tail"] and "ncclReduceScatter" in a["stderr_tail"]
    assert a["error"].startswith("stopped: another rank") and a["seconds"] < 10, a  # rank 0 was not left to sleep out the limit
    assert b["name"] == "ordered_pairs_gloo" and b["ok"]
    # after the measured leg: the native host as a child of rank 0 (`native_host`), the remaining legs as variants
    assert r["native_host"]["value"] == 1.0e12 + 6 and r["native_host"]["host"] == "native"
    v = r["variants"]
    assert v["native_ordered_pairs_copy_one_gpu"]["value"] == 1.0e12 + 5 and v["native_shared_pairs_copy_one_gpu"]["value"] == 1.0e12 + 5
    assert v["native_ordered_pairs_host_one_gpu"]["value"] == 1.0e12 + 5


def test_leg_lists_for_the_requests_a_user_can_make():
    """Which legs, in which order, for which command line — no process is started."""
    import argparse
    bench = _bench()

    def names(legs):
        return [(lg["name"], "skip" in lg) for lg in legs]

    def A(**kw):
        d = dict(gpus=8, exchange=None, ordered_pairs=False, overlap=False, backend="nccl", single_device=False)
        d.update(kw)
        return argparse.Namespace(**d)

    # the default on a full node: shared pairs over RCCL first, north_star's scheme second, then no RCCL, then no peer-to-peer
    assert names(bench.native_legs(A(), 8)) == [("shared_pairs_rccl", False), ("ordered_pairs_rccl", False), ("shared_pairs_copy", False),
                                                ("ordered_pairs_copy", False), ("ordered_pairs_host", False)]
    # a request moves to the front; nothing is tried twice
    assert [n for n, _ in names(bench.native_legs(A(exchange="copy", ordered_pairs=True), 8))] == \
        ["ordered_pairs_copy", "shared_pairs_rccl", "ordered_pairs_rccl", "shared_pairs_copy", "ordered_pairs_host"]
    # the two-phase step runs the ordered-pair kernel and is not host-staged
    assert [n for n, _ in names(bench.native_legs(A(overlap=True), 8))] == ["ordered_pairs_rccl_overlap", "ordered_pairs_copy_overlap"]
    # fewer GPUs than ranks: the request runs (and fails by itself), the other distinct-device legs are skipped, the rehearsal forms follow
    assert names(bench.native_legs(A(), 1)) == [("shared_pairs_rccl", False), ("ordered_pairs_rccl", True), ("shared_pairs_copy", True),
                                                ("ordered_pairs_copy", True), ("ordered_pairs_host", True),
                                                ("shared_pairs_copy_one_gpu", False), ("ordered_pairs_copy_one_gpu", False),
                                                ("ordered_pairs_host_one_gpu", False)]
    assert [n for n, _ in names(bench.native_legs(A(exchange="host-one-gpu"), 1))] == ["ordered_pairs_host_one_gpu", "shared_pairs_copy_one_gpu",
                                                                                     "ordered_pairs_copy_one_gpu"]
    # under the launcher: the torch host's two forms, then the native host without RCCL run by rank 0
    t = bench.torch_legs(A(), 8, 8)
    assert [(lg["name"], lg["host"]) for lg in t] == [("shared_pairs_nccl", "torch"), ("ordered_pairs_nccl", "torch"),
                                                      ("native_shared_pairs_copy", "native"), ("native_ordered_pairs_copy", "native"),
                                                      ("native_ordered_pairs_host", "native")]
    assert [lg["name"] for lg in bench.torch_legs(A(exchange="ring"), 8, 8)] == ["ordered_pairs_nccl_ring"]
    assert [lg["name"] for lg in bench.torch_legs(A(backend="gloo", single_device=True, gpus=2), 2, 1)] == \
        ["shared_pairs_gloo", "ordered_pairs_gloo", "native_shared_pairs_copy_one_gpu", "native_ordered_pairs_copy_one_gpu",
         "native_ordered_pairs_host_one_gpu"]
    import pytest
    with pytest.raises(SystemExit):
        bench.native_legs(A(exchange="ring"), 8)
    with pytest.raises(SystemExit):
        bench.native_legs(A(exchange="host", overlap=True), 8)


def test_power_sampler_reads_the_hwmon_files_of_the_card_with_this_bus_id(tmp_path):
    """bench.py's `roofline.power`: the amdgpu hwmon files of the GPU with the line's PCI bus id, found through the card's
    device symlink; another card's files are not read; a box without them (or with unreadable ones) yields None, never an error."""
    import time

    import bench
    for k, (bus, watts, mhz) in enumerate([("0000:05:00.0", 700, 2100), ("0000:85:00.0", 1300, 2280)]):
        dev = tmp_path / "pci" / bus
        hw = dev / "hwmon" / f"hwmon{k}"
        hw.mkdir(parents=True)
        (hw / "power1_input").write_text(f"{watts * 1000000}\n")
        (hw / "power1_cap").write_text("1400000000\n")
        (hw / "freq1_input").write_text(f"{mhz * 1000000}\n")
        card = tmp_path / "drm" / f"card{k}"
        card.mkdir(parents=True)
        os.symlink(dev, card / "device")
    smp = bench.PowerSampler("0000:85:00.0", period=0.01, root=str(tmp_path / "drm")).start()
    time.sleep(0.1)
    r = smp.stop()
    assert r["samples"] >= 2 and r["mean_w"] == r["max_w"] == 1300.0 and r["cap_w"] == 1400.0
    assert r["sclk_mhz_mean"] == r["sclk_mhz_min"] == 2280.0
    assert bench.PowerSampler("0000:99:00.0", root=str(tmp_path / "drm")).start().stop() is None  # no such card
    (tmp_path / "pci" / "0000:05:00.0" / "hwmon" / "hwmon0" / "power1_input").write_text("garbage\n")
    assert bench.PowerSampler("0000:05:00.0", period=0.01, root=str(tmp_path / "drm")).start().stop() is None  # unreadable: no samples
