"""The whole-program driver nb_solve and the scenario engines behind it: every scenario (P1, P2, one Problem-3 run per
device) from step 0 in one launch stream per GPU; several GPUs; the cheapest-first queue for scenarios beyond one
stream (hw5.cu:490-493,574-596); the batched persistent engine; the ABI's refusal paths added with them."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, case_path, read_golden

pytestmark = pytest.mark.gpu


def _solve_case(nb, oracle, case, **kw):
    s = oracle.read_input(case_path(case, "in"))
    got = nb.capi.solve(s.n, s.planet, s.asteroid, s.q, s.v, s.m, s.is_device, **kw)
    gold = read_golden(case)
    return ("%.16e\n%d\n%d %.16e\n" % got), gold[4]


@pytest.mark.parametrize("case", ["b30", "b90", "b200", "b512", "b1024"])
def test_solve_spread_over_two_device_slots(nb, oracle, case):
    """devices=[0,0]: the multi-GPU code path (one host thread and launch stream per listed device, scenarios dealt
    round-robin: P1 -> first, P2 -> second, ...) on the one GPU of the box.  b512 / b1024 take the split layout (n > 256:
    a stream per scenario, two follower streams per device slot).  Answers byte-identical to the goldens."""
    text, gold = _solve_case(nb, oracle, case, devices=[0, 0])
    assert text == gold


@pytest.mark.parametrize("case,streams", [("b200", "merged"), ("b200", "split"), ("b512", "split"), ("b1024", "split")])
def test_cross_gpu_handoff_of_the_arrival_snapshot(nb, case, streams, tmp_path):
    """hw5.cu:482-484 uploads P2's arrival snapshot on the OTHER GPU before a Problem-3 run starts there.  Here that is
    activate_follower's host-staged branch; with devices = 0,0 both slots are one physical GPU, so the hand-off is forced
    through host memory for runs that sit on another device slot than P2 (nb_solve_options.handoff, NB_SOLVE_HANDOFF=host
    in the CLI).  The trace must show the branch was taken, and the three lines must still be the golden ones."""
    out = str(tmp_path / "out")
    env = dict(os.environ, NB_DEVICES="0,0", NB_SOLVE_HANDOFF="host", NB_SOLVE_TRACE="1", NB_SOLVE_STREAMS=streams,
               NB_SOLVE_P3_PARALLEL="16")  # every device's run starts, whatever its rank in the queue
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), case_path(case, "in"), out], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    starts = [ln for ln in p.stderr.splitlines() if ln.startswith("[followers]")]
    assert any("host-staged" in ln for ln in starts), p.stderr   # P2 is on slot 1; devices 0, 2, ... run on slot 0
    assert any("device copy" in ln for ln in starts) or len(starts) == 1, p.stderr
    assert open(out).read() == read_golden(case)[4]
    # and without the hook nothing is staged on a one-GPU box
    env.pop("NB_SOLVE_HANDOFF")
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), case_path(case, "in"), out], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0 and "host-staged" not in p.stderr and "[followers]" in p.stderr
    assert open(out).read() == read_golden(case)[4]


def test_solve_options_are_validated(nb, oracle):
    c = nb.capi
    s = oracle.read_input(case_path("b20", "in"))
    for kw in (dict(max_batch=1), dict(max_batch=9), dict(graph_chunk=3), dict(graph_chunk=5000), dict(p3_parallel=-1),
               dict(handoff=7)):
        with pytest.raises(c.NBodyError) as e:
            c.solve(s.n, s.planet, s.asteroid, s.q, s.v, s.m, s.is_device, **kw)
        assert e.value.code == c.NB_ERR_INVALID and "nb_solve_options" in str(e.value)
    big = oracle.read_input(case_path("b200", "in"))
    with pytest.raises(c.NBodyError) as e:
        c.solve(big.n, big.planet, big.asteroid, big.q, big.v, big.m, big.is_device, engine="persistent")
    assert e.value.code == c.NB_ERR_INVALID


@pytest.mark.parametrize("streams,p3", [("merged", "1"), ("split", "1"), ("split", "16")])
@pytest.mark.parametrize("case", ["b200", "b512"])
def test_solve_stream_layouts_agree(nb, oracle, case, streams, p3):
    """The per-step engine's two stream layouts (one shared graph per GPU / a stream per scenario) and both Problem-3
    policies (queued cheapest-first / all at once) give the golden answers."""
    text, gold = _solve_case(nb, oracle, case, streams=streams, p3_parallel=int(p3))
    assert text == gold


@pytest.mark.parametrize("case,cap", [("b30", 2), ("b80", 3), ("b200", 4)])
def test_solve_queue_beyond_one_stream(nb, oracle, case, cap):
    """nb_solve_options.max_batch < 2 + D: the devices that do not fit the first wave are queued in ascending arrival step
    (hw5.cu:574-585) and skipped once they cannot beat a feasible one (hw5.cu:490-493); the answer does not change.
    b80: device 76 (arrival 151213) is the answer; 77-79 arrive later."""
    text, gold = _solve_case(nb, oracle, case, max_batch=cap)
    assert text == gold


def test_batched_persistent_engine_equals_single_runs(nb, oracle):
    """K3 batched (one launch, one workgroup per scenario) vs one persistent launch per scenario on b50: P1, P2 with
    snapshots, and both Problem-3 runs from step 0 — identical results and bitwise identical final states."""
    c = nb.capi
    s = oracle.read_input(case_path("b50", "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    m0 = s.m.copy()
    m0[devs] = 0.0
    kws = [dict(kind=c.NB_SCN_MIN_DIST, planet=s.planet, asteroid=s.asteroid, last_step=60000),
           dict(kind=c.NB_SCN_FIRST_HIT, planet=s.planet, asteroid=s.asteroid, watch=devs),
           dict(kind=c.NB_SCN_MISSILE, planet=s.planet, asteroid=s.asteroid, watch=[devs[0]]),
           dict(kind=c.NB_SCN_MISSILE, planet=s.planet, asteroid=s.asteroid, watch=[devs[1]], last_step=120000)]
    out = {}
    for mode in ("single", "batched"):
        ctxs = [c.Context(s.n) for _ in kws]
        for k, x in enumerate(ctxs):
            x.set_state(s.q, s.v, m0 if k == 0 else s.m, s.is_device)
        res = [x.run_scenario(engine=2, **kw) for x, kw in zip(ctxs, kws)] if mode == "single" else \
            c.run_scenarios_batched(ctxs, [dict(engine=2, **kw) for kw in kws])
        snap = []
        for slot in range(len(devs)):  # P2's snapshots must be the same states
            with c.Context(s.n) as r:
                r.restore_snapshot_from(ctxs[1], slot)
                snap.append(r.get_state())
        out[mode] = (res, [x.get_state() for x in ctxs], snap)
        for x in ctxs:
            x.close()
    (ra, sa, na), (rb, sb, nb_) = out["single"], out["batched"]
    assert ra == rb, (ra, rb)
    assert rb[1]["hit_step"] == 103140 and rb[1]["arrival_step"] == [87218, 89015]
    assert rb[2]["hit_step"] == -2 and rb[2]["arrival_step"] == [87218] and rb[3]["steps_done"] == 120000
    assert rb[0]["steps_done"] == 60000
    for (qa, va), (qb, vb) in list(zip(sa, sb))[:1] + list(zip(sa, sb))[2:] + list(zip(na, nb_)):
        assert np.array_equal(qa, qb) and np.array_equal(va, vb)  # (P2's own final state is unspecified after a hit)


def test_missile_scenario_takes_one_device(nb, oracle):
    """One device is destroyed per Problem-3 run (hw5.cu:289-309): more than one watched device is refused by both
    entry points instead of silently destroying only one of them."""
    c = nb.capi
    s = oracle.read_input(case_path("b80", "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    with c.Context(s.n) as a, c.Context(s.n) as b:
        for x in (a, b):
            x.set_state(s.q, s.v, s.m, s.is_device)
        with pytest.raises(c.NBodyError) as e:
            a.run_scenario(c.NB_SCN_MISSILE, s.planet, s.asteroid, watch=devs[:2], last_step=10)
        assert e.value.code == c.NB_ERR_INVALID
        with pytest.raises(c.NBodyError) as e:
            c.run_scenarios_batched([a, b], [dict(kind=c.NB_SCN_MISSILE, planet=s.planet, asteroid=s.asteroid,
                                                  watch=devs[:2], last_step=10)] * 2)
        assert e.value.code == c.NB_ERR_INVALID
        r = a.run_scenario(c.NB_SCN_MISSILE, s.planet, s.asteroid, watch=devs[:1], last_step=10)  # one is fine
        assert r["steps_done"] == 10


def test_restore_snapshot_of_a_device_that_never_arrived(nb, oracle):
    """A FIRST_HIT run that ends before device k's missile arrives leaves slot k unwritten: restoring it is a call
    sequence error (NB_ERR_STATE), not an upload of uninitialised memory; NB_SCN_NO_SNAPSHOT keeps no slots at all."""
    c = nb.capi
    s = oracle.read_input(case_path("b30", "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    with c.Context(s.n) as p2, c.Context(s.n) as dst:
        p2.set_state(s.q, s.v, s.m, s.is_device)
        r = p2.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs, last_step=1000)
        assert r["arrival_step"] == [-2, -2] and r["hit_step"] == -2
        with pytest.raises(c.NBodyError) as e:
            dst.restore_snapshot_from(p2, 0)
        assert e.value.code == c.NB_ERR_STATE and "no missile arrival" in str(e.value)
    with c.Context(s.n) as p2, c.Context(s.n) as dst:
        p2.set_state(s.q, s.v, s.m, s.is_device)
        r = p2.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs, flags=c.NB_SCN_NO_SNAPSHOT)
        assert r["arrival_step"] == [178526, 177846] and r["hit_step"] == 180769
        with pytest.raises(c.NBodyError):
            dst.restore_snapshot_from(p2, 1)


def test_load_state_refuses_other_parameters(nb, oracle, tmp_path):
    """A checkpoint resumes the run it came from: another dt, eps, G or precision is NB_ERR_INVALID with the reason in
    nb_last_error; the explicit route (read_state_file + set_state) still loads it."""
    c = nb.capi
    s = oracle.read_input(case_path("b20", "in"))
    path = str(tmp_path / "b20.nbst")
    with c.Context(s.n) as ctx:
        ctx.set_state(s.q, s.v, s.m, s.is_device)
        ctx.step(1, 3)
        ctx.save_state(path, step=3)
        q3, v3 = ctx.get_state()
    for kw in (dict(dt=30.0), dict(eps=2e-3), dict(G=1e-11)):
        with c.Context(s.n, **kw) as ctx, pytest.raises(c.NBodyError) as e:
            ctx.load_state(path)
        assert e.value.code == c.NB_ERR_INVALID and "does not match" in str(e.value)
    with c.Context(s.n, c.NB_F32_ACC64) as ctx, pytest.raises(c.NBodyError) as e:
        ctx.load_state(path)
    assert e.value.code == c.NB_ERR_INVALID
    h, q, v, m, dev = c.read_state_file(path)
    assert h["step"] == 3 and np.array_equal(q, q3) and np.array_equal(v, v3) and np.array_equal(dev, s.is_device)
    with c.Context(s.n, dt=30.0) as ctx:
        ctx.set_state(q, v, m, dev)
        ctx.step(4, 1)


@pytest.mark.parametrize("case", ["b30", "b200"])
def test_cli_binary_input(nb, case, tmp_path):
    """hw5 <input> accepts the binary (NBODYST2) form of the same data — nbconv's output — and prints
    the same three lines as for the text file."""
    st, out = str(tmp_path / f"{case}.nbst"), str(tmp_path / "out")
    subprocess.run([os.path.join(ROOT, "bin", "nbconv"), case_path(case, "in"), st], check=True)
    subprocess.run([os.path.join(ROOT, "bin", "hw5"), st, out], check=True, timeout=300)
    assert open(out).read() == read_golden(case)[4]
    # a checkpoint without planet/asteroid is not a program input
    c = nb.capi
    _, q, v, m, dev = c.read_state_file(st)
    plain = str(tmp_path / "plain.nbst")
    c.write_state_file(plain, q, v, m, dev)
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), plain, out], capture_output=True)
    assert p.returncode == 1 and b"cannot read state file" in p.stderr and b"planet/asteroid" in p.stderr
    # nor is a mid-run state, an fp32 one, or one written under other constants: solved as a step-0 fp64 input under
    # param::'s values its three lines would be wrong without a word
    h = c.read_state_file(st)[0]
    for kw, word in ((dict(step=7), b"step != 0"), (dict(precision=c.NB_F32), b"precision"), (dict(dt=30.0), b"dt differs"),
                     (dict(eps=1e-2), b"eps differs"), (dict(G=1.0), b"G differs")):
        odd = str(tmp_path / "odd.nbst")
        c.write_state_file(odd, q, v, m, dev, planet=h["planet"], asteroid=h["asteroid"], **kw)
        p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), odd, out], capture_output=True)
        assert p.returncode == 1 and word in p.stderr, (kw, p.stderr)


@pytest.mark.parametrize("case,last", [("b200", 9013), ("b200", 24000), ("b512", 5001)])
def test_graph_replay_equals_eager_launches(nb, oracle, case, last):
    """The per-step engine replays a captured hipGraph of 1000 launches for long runs (step index from a device control
    word, |sin| from the table, tail handled in-kernel) — same kernel body as the eager launches (NB_SCN_EAGER): results,
    final states and arrival snapshots must be identical bit for bit, for ranges that are not whole replays and leave
    the state in either ping-pong buffer."""
    c = nb.capi
    s = oracle.read_input(case_path(case, "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    m0 = s.m.copy()
    m0[devs] = 0.0
    outs = []
    for flags in (c.NB_SCN_EAGER, 0):
        with c.Context(s.n) as p1, c.Context(s.n) as p2:
            p1.set_state(s.q, s.v, m0, s.is_device)
            p2.set_state(s.q, s.v, s.m, s.is_device)
            r1 = p1.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, first_step=0, last_step=last, engine=1, flags=flags)
            r2 = p2.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs, last_step=last, engine=1, flags=flags)
            snaps = []
            for k, a in enumerate(r2["arrival_step"]):
                if a >= 0:
                    with c.Context(s.n) as x:
                        x.restore_snapshot_from(p2, k)
                        snaps.append(x.get_state())
            # and on from there: a second run continues from the state the first one left
            r1b = p1.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, first_step=last, last_step=last + 4500, engine=1,
                                  flags=flags)
            outs.append((r1, r2, r1b, p1.get_state(), p2.get_state(), snaps))
    a, b = outs
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], (a[:3], b[:3])
    assert a[0]["steps_done"] == last and a[2]["steps_done"] == last + 4500
    for (qa, va), (qb, vb) in [(a[3], b[3]), (a[4], b[4])] + list(zip(a[5], b[5])):
        assert np.array_equal(qa, qb) and np.array_equal(va, vb)
    if case == "b200" and last == 24000:
        assert a[1]["arrival_step"][0] == 19248 and len(a[5]) >= 1


def test_graph_replay_stops_at_the_hit(nb, oracle):
    """Full Problem 2 of b200 through the replayed graph: the hit ends the scenario (steps_done = hit step), arrivals
    before it are recorded (hit 102281, device 197 reached at 19248); a batch of a MIN_DIST and a
    MISSILE scenario with different ranges shares one graph."""
    c = nb.capi
    s = oracle.read_input(case_path("b200", "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    with c.Context(s.n) as p2:
        p2.set_state(s.q, s.v, s.m, s.is_device)
        r = p2.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs)
    assert r["hit_step"] == 102281 and r["steps_done"] == 102281
    assert r["arrival_step"][0] == 19248 and all(a == -2 or 0 < a < 102281 for a in r["arrival_step"])
    with c.Context(s.n) as a, c.Context(s.n) as b:
        a.set_state(s.q, s.v, s.m, s.is_device)
        b.set_state(s.q, s.v, s.m, s.is_device)
        ra, rb = c.run_scenarios_batched([a, b], [
            dict(kind=c.NB_SCN_MIN_DIST, planet=s.planet, asteroid=s.asteroid, last_step=4321),
            dict(kind=c.NB_SCN_MISSILE, planet=s.planet, asteroid=s.asteroid, watch=[devs[0]], last_step=30001)])
        qa, _ = a.get_state()
    assert ra["steps_done"] == 4321 and rb["steps_done"] == 30001 and rb["arrival_step"] == [19248] and rb["hit_step"] == -2
    ref = s.copy()
    oracle.run_steps(ref, 1, 4321, omp=True)
    assert np.all(np.abs(qa - ref.q) <= 1e-9 * np.abs(ref.q).max())


def test_cli_with_a_shorter_graph_chunk(nb, tmp_path):
    """NB_GRAPH_CHUNK=100: ten times as many, shorter replays (what profiling under rocprofv3 needs) — same three lines.
    NB_HW5_CLEAN_EXIT=1 makes hw5 return from main instead of _Exit, so this run also exercises the process teardown
    (HIP exit handlers after graphs, borrowed streams, destruction order of the contexts)."""
    out = str(tmp_path / "out")
    env = dict(os.environ, NB_GRAPH_CHUNK="100", NB_HW5_CLEAN_EXIT="1", NB_SOLVE_TRACE="1")
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), case_path("b200", "in"), out], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    assert "x 100 launches" in p.stderr  # the option reached the graph capture
    assert open(out).read() == read_golden("b200")[4]


@pytest.mark.parametrize("case", ["b20", "b1024"])
def test_cli_clean_exit_runs_the_teardown(nb, case, tmp_path):
    """Both engines (persistent for b20, replayed graphs on a stream per scenario for b1024) with a normal return from
    main: a fault in the teardown path would show as a non-zero exit status."""
    out = str(tmp_path / "out")
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), case_path(case, "in"), out], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, NB_HW5_CLEAN_EXIT="1"))
    assert p.returncode == 0, (p.returncode, p.stderr)
    assert open(out).read() == read_golden(case)[4]


def test_graph_chunk_of_a_single_scenario(nb, oracle):
    """nb_scenario.graph_chunk: the same run through graphs of 1000 (default), 100 and 2 launches — identical results
    and final state bits; an odd or oversized chunk is refused."""
    c = nb.capi
    s = oracle.read_input(case_path("b200", "in"))
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    outs = []
    for chunk in (0, 100, 2):
        with c.Context(s.n) as x:
            x.set_state(s.q, s.v, s.m, s.is_device)
            r = x.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs, last_step=4321 if chunk != 2 else 4001,
                               engine=1, graph_chunk=chunk)
            outs.append((r, x.get_state()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1][0], outs[1][1][0])
    assert outs[2][0]["steps_done"] == 4001
    with c.Context(s.n) as x:
        x.set_state(s.q, s.v, s.m, s.is_device)
        for bad in (3, 4002, -2):
            with pytest.raises(c.NBodyError) as e:
                x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=5000, engine=1, graph_chunk=bad)
            assert e.value.code == c.NB_ERR_INVALID


def test_step_stamps_measure_without_changing_results(nb, oracle):
    """nb_enable_step_stamps: every step launch of the per-step engine records the GPU wall clock at entry and after its
    last store — entry < exit, launches in order — and the trajectory is bit for bit that of the unstamped run, for eager
    launches and for a replayed graph."""
    c = nb.capi
    s = oracle.read_input(case_path("b200", "in"))
    with c.Context(s.n) as x, pytest.raises(c.NBodyError) as e:  # the product build carries no instrumentation
        x.enable_step_stamps(8)
    assert e.value.code == c.NB_ERR_STATE and "libnbody_amd_stamps.so" in str(e.value)
    with c.use_library(c.stamps_library_path()):
        _stamps_checks(c, s)


def _stamps_checks(c, s):
    for flags, last in ((c.NB_SCN_EAGER, 64), (0, 64 * 65 - 2)):  # the last replay: 62 steps, the final monitor, one idle node
        out = []
        for slots in (0, 64):
            with c.Context(s.n) as x:
                x.set_state(s.q, s.v, s.m, s.is_device)
                if slots:
                    x.enable_step_stamps(slots)
                r = x.run_scenario(c.NB_SCN_MIN_DIST, s.planet, s.asteroid, last_step=last, engine=1, flags=flags,
                                   graph_chunk=64)
                st = x.read_step_stamps(slots) if slots else None
                out.append((r, x.get_state(), st))
                if slots:
                    with pytest.raises(c.NBodyError):
                        x.read_step_stamps(slots + 1)
                    x.enable_step_stamps(0)  # off again
                    with pytest.raises(c.NBodyError):
                        x.read_step_stamps(1)
        (r0, (q0, v0), _), (r1, (q1, v1), st) = out
        assert r0 == r1 and np.array_equal(q0, q1) and np.array_equal(v0, v1)
        st = st.astype(np.int64)
        done = st[:, 1] > 0
        assert done.sum() >= 60                       # (the monitor-only launch after the last step leaves no exit stamp)
        dur = (st[:, 1] - st[:, 0])[done]
        assert np.all(dur > 0) and np.all(dur < 100000)   # 0 < duration < 1 ms in 10 ns ticks
        k = np.flatnonzero(done[:-1] & done[1:])
        if not flags:  # slots of one replay are consecutive nodes of the graph
            assert np.all(st[k + 1, 0] >= st[k, 1])       # a node starts after its predecessor's last store
    with c.Context(1024, c.NB_F32, eps=1e-3) as x, pytest.raises(c.NBodyError):
        x.enable_step_stamps(8)  # fp64 engine only


@pytest.mark.parametrize("case,devices,handoff", [("b30", "0", ""), ("b200", "0,0", "host"), ("b512", "0,0", "host"),
                                                  ("b1024", "0", "")])
def test_whole_program_under_host_asan(nb, case, devices, handoff, tmp_path):
    """bin/asan/hw5 (`make asan`): the product's host code — nb_solve's threads, the graph scheduler and follower queue, the
    host-staged hand-off, the I/O — compiled with AddressSanitizer + UBSan (device code: the plain gfx950 build) and run on the GPU.
    No report, golden output.  (GPU-side sanitizers are not available on the pool; the reference's equivalent was
    cuda-memcheck, hw5.cu:631-642.)"""
    exe = os.path.join(ROOT, "bin", "asan", "hw5")
    if not os.path.exists(exe):
        pytest.skip("bin/asan/hw5 not built (make asan)")
    out = str(tmp_path / "out")
    env = dict(os.environ, NB_DEVICES=devices, NB_HW5_CLEAN_EXIT="1", ASAN_OPTIONS="detect_leaks=0")
    if handoff:
        env["NB_SOLVE_HANDOFF"] = handoff
    p = subprocess.run([exe, case_path(case, "in"), out], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-2000:]
    assert open(out).read() == read_golden(case)[4]


@pytest.mark.parametrize("case,devices,handoff", [("b30", "0,0", ""), ("b60", "0,0,0", ""), ("b200", "0,0", "host")])
def test_whole_program_under_host_tsan(nb, case, devices, handoff, tmp_path):
    """bin/tsan/hw5 (`make tsan`): the same program with ThreadSanitizer on the host code (the reference's known races —
    the unsynchronised read of gravity_device_id, hw5.cu:491, and `cost < missile_cost` tested outside the lock, hw5.cu:512 — are
    what this looks for in OUR host: nb_solve drives every device slot from its own host thread).  The ROCm runtime is not
    instrumented; what TSan sees inside libamdhip64 / libhsa-runtime64 is suppressed (bench/tsan.supp), anything located in
    libnbody_amd.so or hw5 is a failure.  Golden output."""
    exe = os.path.join(ROOT, "bin", "tsan", "hw5")
    if not os.path.exists(exe):
        pytest.skip("bin/tsan/hw5 not built (make tsan)")
    out = str(tmp_path / "out")
    supp = os.path.join(ROOT, "bench", "tsan.supp")
    env = dict(os.environ, NB_DEVICES=devices, NB_HW5_CLEAN_EXIT="1",
               TSAN_OPTIONS=f"suppressions={supp} halt_on_error=0 report_signal_unsafe=0 exitcode=0")
    if handoff:
        env["NB_SOLVE_HANDOFF"] = handoff
    p = subprocess.run([exe, case_path(case, "in"), out], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    ours = [ln for ln in p.stderr.splitlines() if ln.startswith("SUMMARY: ThreadSanitizer") and
            ("libnbody_amd" in ln or "/hw5" in ln or "nbody_" in ln)]
    assert not ours, "\n".join(ours)
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-3000:]   # (with the runtime's own reports suppressed: none at all)
    assert open(out).read() == read_golden(case)[4]
