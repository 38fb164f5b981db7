"""Edge cases of the scenario drivers (P1/P2/P3) on hand-made systems, GPU (nb_solve / hw5) vs the oracle:
no hit, hit at step 0, hit with no devices, a device whose missile arrives but cannot prevent the hit, coincident
bodies, massless bodies — and the ABI's error paths."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _system(oracle, bodies, planet=0, asteroid=1):
    """bodies: list of (q, v, m, is_device)."""
    s = oracle.System(len(bodies), planet, asteroid)
    for i, (q, v, m, d) in enumerate(bodies):
        s.q[:, i], s.v[:, i], s.m[i], s.is_device[i] = q, v, m, d
    return s


def _write_in(s, path):
    with open(path, "w") as f:
        f.write(f"{s.n} {s.planet} {s.asteroid}\n")
        for i in range(s.n):
            vals = list(s.q[:, i]) + list(s.v[:, i]) + [s.m[i]]
            f.write(" ".join("%.16e" % x for x in vals) + (" device\n" if s.is_device[i] else " rock\n"))


PLANET = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 6e24, 0)
CASES = {
    # asteroid on a collision course; a light device nearby: its missile arrives, the hit happens anyway
    "collision_device_useless": [PLANET, ((3e8, 0, 0), (-3e3, 0, 0), 1e12, 0), ((1e8, 1e8, 0), (0, 0, 0), 1e15, 1)],
    # asteroid leaving: never hits -> -2 / -1 0
    "no_hit": [PLANET, ((3e8, 0, 0), (5e4, 0, 0), 1e12, 0), ((1e8, 1e8, 0), (0, 0, 0), 1e15, 1)],
    # already inside the planet radius at step 0
    "hit_at_step_0": [PLANET, ((5e6, 0, 0), (0, 0, 0), 1e12, 0), ((1e9, 1e9, 0), (0, 0, 0), 1e15, 1)],
    # hit, but the input has no device at all
    "hit_no_devices": [PLANET, ((3e8, 0, 0), (-3e3, 0, 0), 1e12, 0), ((0, 5e8, 0), (1e3, 0, 0), 1e20, 0)],
    # two bodies at the same place (r = 0: only the softening keeps the pair finite) and a massless one
    "coincident_and_massless": [PLANET, ((3e8, 0, 0), (-3e3, 0, 0), 1e12, 0), ((0, 7e8, 0), (0, 0, 0), 1e18, 0),
                                ((0, 7e8, 0), (0, 0, 0), 1e18, 0), ((1e9, 0, 4e8), (0, 10, 0), 0.0, 0),
                                ((2e8, -1e8, 0), (0, 0, 0), 1e16, 1)],
    # the light device next to the planet is reached first (step 2) but destroying it changes nothing; the heavy one
    # (step 4) is what pulls the asteroid in: the cheapest-first queue must hand over after the first run fails
    "first_arrival_useless_second_saves": [PLANET, ((4e8, 6e7, 0), (-2e3, 0, 0), 1e12, 0), ((5e7, 5e7, 0), (0, 0, 0), 1e10, 1),
                                           ((1e8, -2e8, 0), (0, 0, 0), 3e24, 1)],
    # six devices (more than the four whose monitor state the kernels keep in registers; 2 + 6 scenarios = one full batch):
    # five light bystanders, one of which is reached at the same step (4) as the heavy culprit
    "six_devices_with_a_tie": [PLANET, ((4e8, 6e7, 0), (-2e3, 0, 0), 1e12, 0), ((5e7, 5e7, 0), (0, 0, 0), 1e10, 1),
                               ((-2e8, 1e8, 0), (0, 0, 0), 1e11, 1), ((3e8, -3e8, 1e8), (0, 0, 0), 1e9, 1),
                               ((-4e8, -1e8, 0), (0, 0, 0), 1e10, 1), ((1e8, -2e8, 0), (0, 0, 0), 3e24, 1),
                               ((6e8, 6e8, 0), (0, 0, 0), 1e10, 1)],
    # a heavy device that pulls the asteroid into the planet: destroying it in time avoids the hit
    "device_causes_hit": [PLANET, ((4e8, 6e7, 0), (-2e3, 0, 0), 1e12, 0), ((1e8, -2e8, 0), (0, 0, 0), 3e24, 1),
                          ((-9e8, 9e8, 0), (0, 0, 0), 1e10, 1)],
}


@pytest.mark.parametrize("engine", ["persistent", "steps", "steps-all-at-once"])
@pytest.mark.parametrize("name", list(CASES))
def test_solve_matches_oracle(nb, oracle, name, engine, tmp_path):
    """engine: the whole program through the persistent engine (what these small systems get by default), through the
    per-step engine (graph replay, a stream per scenario, Problem-3 runs queued cheapest-first), and through it with every
    Problem-3 run started as soon as its missile arrives."""
    opts = dict(engine=engine.split("-")[0], p3_parallel=16 if engine.endswith("all-at-once") else 0)
    s = _system(oracle, CASES[name])
    ref_min = oracle.problem1(s)
    ref, details = oracle.problem23(s)
    got = nb.capi.solve(s.n, s.planet, s.asteroid, s.q, s.v, s.m, s.is_device, **opts)
    assert got[1] == ref.hit_time_step and got[2] == ref.gravity_device_id, (name, got, details)
    assert got[3] == ref.missile_cost
    assert abs(got[0] - ref_min) <= 1e-9 * ref_min
    # and through the CLI, against the oracle's CLI
    inp, out, out_ref = tmp_path / "c.in", tmp_path / "c.out", tmp_path / "c.ref"
    _write_in(s, inp)
    env = dict(os.environ, NB_SOLVE_ENGINE=opts["engine"], NB_SOLVE_P3_PARALLEL=str(opts["p3_parallel"]))  # hw5 maps them
    subprocess.run([os.path.join(ROOT, "bin", "hw5"), str(inp), str(out)], check=True, timeout=300, env=env)
    oracle.solve_file(str(inp), str(out_ref))
    a, b = out.read_text().split("\n"), out_ref.read_text().split("\n")
    assert a[1:] == b[1:], (name, a, b)            # hit step, device id and cost: exactly
    assert abs(float(a[0]) - float(b[0])) <= 1e-9 * float(b[0])


@pytest.mark.parametrize("cap", [2, 3, 4])
def test_arrival_tie_in_the_queued_wave(nb, oracle, cap):
    """six_devices_with_a_tie through the persistent engine with max_batch below 2 + D: the devices beyond the first
    wave are queued by arrival step, and the two that arrive at step 4 (one useless, one feasible) must resolve exactly
    as in the oracle and in the per-step engine — ties keep index order (stable sort, strict < in the selection)."""
    s = _system(oracle, CASES["six_devices_with_a_tie"])
    ref, _ = oracle.problem23(s)
    got = nb.capi.solve(s.n, s.planet, s.asteroid, s.q, s.v, s.m, s.is_device, engine="persistent", max_batch=cap)
    assert (got[1], got[2], got[3]) == (ref.hit_time_step, ref.gravity_device_id, ref.missile_cost)
    steps = nb.capi.solve(s.n, s.planet, s.asteroid, s.q, s.v, s.m, s.is_device, engine="steps")
    assert steps[1:] == got[1:]


def test_expected_shapes_of_the_edge_cases(oracle):
    """Make sure the hand-made systems really exercise what their names say (guards against a vacuous test)."""
    res = {k: oracle.problem23(_system(oracle, v), max_detail=16) for k, v in CASES.items()}
    assert res["no_hit"][0].hit_time_step == -2
    assert res["hit_at_step_0"][0].hit_time_step == 0
    assert res["hit_no_devices"][0].hit_time_step > 0 and res["hit_no_devices"][0].gravity_device_id == -1
    r, d = res["collision_device_useless"]
    assert r.hit_time_step > 0 and d[0]["arrival_step"] >= 0 and not d[0]["feasible"] and r.gravity_device_id == -1
    r, d = res["device_causes_hit"]
    assert r.hit_time_step > 0 and d[0]["feasible"] and r.gravity_device_id == 2 and r.missile_cost > 0
    assert d[1]["arrival_step"] > d[0]["arrival_step"] and not d[1]["feasible"]  # the far, light device cannot help
    r, d = res["six_devices_with_a_tie"]
    assert len(d) == 6 and [x["arrival_step"] for x in d] == [2, 4, 8, 7, 4, 15] and [x["feasible"] for x in d] == \
        [False, False, False, False, True, False] and r.gravity_device_id == 6
    r, d = res["first_arrival_useless_second_saves"]
    assert d[0]["arrival_step"] < d[1]["arrival_step"] < r.hit_time_step and not d[0]["feasible"] and d[1]["feasible"]
    assert r.gravity_device_id == 3 and r.missile_cost == d[1]["cost"]


def test_abi_error_paths(nb):
    c = nb.capi
    with c.Context(8) as ctx:
        with pytest.raises(c.NBodyError) as e:
            ctx.step(1, 1)                       # no state yet
        assert e.value.code == c.NB_ERR_STATE
        z = np.zeros((3, 8)); z[0] = np.arange(8)
        ctx.set_state(z, z, np.ones(8))
        with pytest.raises(c.NBodyError):
            ctx.set_mass(8, 0.0)                 # index out of range
        with pytest.raises(c.NBodyError):
            ctx.run_scenario(c.NB_SCN_MIN_DIST, 0, 9, last_step=10)     # asteroid out of range
        with pytest.raises(c.NBodyError):
            ctx.run_scenario(c.NB_SCN_FIRST_HIT, 0, 1, last_step=10, watch=[99])
        with pytest.raises(c.NBodyError):
            ctx.run_scenario(7, 0, 1, last_step=10)                     # unknown kind
        with pytest.raises(c.NBodyError):
            ctx.run_scenario(c.NB_SCN_MIN_DIST, 0, 1, first_step=5, last_step=4)
        with c.Context(8) as other, pytest.raises(c.NBodyError):
            other.restore_snapshot_from(ctx, 0)  # no snapshot was taken
    with pytest.raises(c.NBodyError) as e:
        c.Context(8, device=64)
    assert e.value.code == c.NB_ERR_NO_DEVICE
    with pytest.raises(c.NBodyError):
        c.Context(0)
    with c.Context(8, c.NB_F32) as f32, pytest.raises(c.NBodyError):
        z = np.zeros((3, 8)); z[0] = np.arange(8)
        f32.set_state(z, z, np.ones(8))
        f32.run_scenario(c.NB_SCN_MIN_DIST, 0, 1, last_step=3)          # scenarios are fp64-only


def test_cli_unreadable_input(nb, tmp_path):
    p = subprocess.run([os.path.join(ROOT, "bin", "hw5"), str(tmp_path / "nope.in"), str(tmp_path / "o")],
                       capture_output=True)
    assert p.returncode == 1 and b"cannot read" in p.stderr


def test_permutation_invariance(nb, oracle):
    """Reordering the bodies (what hw5.cu:110-130 does to put planet/asteroid/devices first) must not change the
    physics: same accelerations for the same bodies, to rounding."""
    from conftest import case_path
    s = oracle.read_input(case_path("b200", "in"))
    perm = np.random.default_rng(3).permutation(s.n)
    with nb.capi.Context(s.n) as a, nb.capi.Context(s.n) as b:
        a.set_state(s.q, s.v, s.m, s.is_device)
        b.set_state(s.q[:, perm], s.v[:, perm], s.m[perm], s.is_device[perm])
        acc_a, acc_b = a.accel(1234), b.accel(1234)
    assert np.all(np.abs(acc_a[:, perm] - acc_b) <= 1e-12 * np.abs(acc_a).max())
    # Newton's third law on the effective masses of that step
    me = oracle.effective_mass(1234, s.m, s.is_device, 60.0)
    p = (acc_a * me).sum(axis=1)
    assert np.all(np.abs(p) <= 1e-10 * (np.abs(acc_a) * me).sum(axis=1))


@pytest.mark.parametrize("engine", [1, 2])
def test_arrival_on_the_final_state_takes_a_complete_snapshot(nb, oracle, engine):
    """The missile reaches the device exactly at last_step: the arrival is found by the final (monitor-only) evaluation,
    whose snapshot must still cover every body."""
    s = _system(oracle, CASES["collision_device_useless"] + [((5e9, 0, 0), (0, 1, 0), 1e10, 0)] * 70)  # 73 bodies
    _, details = oracle.problem23(s)
    arr = details[0]["arrival_step"]
    assert arr == 3
    ref = s.copy()
    oracle.run_steps(ref, 1, arr)
    with nb.capi.Context(s.n) as ctx, nb.capi.Context(s.n) as c3:
        ctx.set_state(s.q, s.v, s.m, s.is_device)
        r = ctx.run_scenario(nb.capi.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=[2], last_step=arr, engine=engine)
        assert r["arrival_step"] == [arr] and r["hit_step"] == -2 and r["steps_done"] == arr
        c3.restore_snapshot_from(ctx, 0)
        q, v = c3.get_state()
    assert np.all(np.abs(q - ref.q) <= 1e-12 * np.abs(ref.q).max()) and np.all(np.abs(v - ref.v) <= 1e-12 * np.abs(ref.v).max())


@pytest.mark.parametrize("seed", range(12))
def test_engines_agree_on_perturbed_systems(nb, oracle, seed):
    """Differential test of the three scenario engines on randomly perturbed copies of the hand-made systems (positions,
    velocities and masses within +-8 %, a random number of far-away filler bodies): eager per-step launches and their
    hipGraph replay must agree bit for bit; the persistent engine must report the same hit / arrival steps (its summation
    split differs, so its states agree to rounding only); and everything must match the oracle's bookkeeping."""
    c = nb.capi
    rng = np.random.default_rng(1000 + seed)
    name = list(CASES)[seed % len(CASES)]
    bodies = []
    for q, v, m, d in CASES[name]:
        f = lambda x: tuple(np.asarray(x, dtype=float) * (1 + 0.08 * (2 * rng.random(3) - 1)))  # noqa: E731
        bodies.append((f(q), f(v), m * (1 + 0.08 * (2 * rng.random() - 1)), d))
    for _ in range(int(rng.integers(0, 40))):  # fillers: light, far, slow
        bodies.append((tuple(5e9 * (2 * rng.random(3) - 1) + 2e10), tuple(10 * rng.random(3)), 1e8 * rng.random(), 0))
    s = _system(oracle, bodies)
    devs = [int(i) for i in np.flatnonzero(s.is_device)]
    last = 9000 + int(rng.integers(0, 7))
    p = oracle.make_params(n_steps=last)
    ref, details = oracle.problem23(s, params=p, max_detail=16)
    runs = {}
    for label, engine, flags in (("eager", 1, c.NB_SCN_EAGER), ("graph", 1, 0), ("persistent", 2, 0)):
        with c.Context(s.n) as x:
            x.set_state(s.q, s.v, s.m, s.is_device)
            r = x.run_scenario(c.NB_SCN_FIRST_HIT, s.planet, s.asteroid, watch=devs, last_step=last, engine=engine, flags=flags)
            runs[label] = (r, x.get_state())
    (re_, se), (rg, sg), (rp, sp) = runs["eager"], runs["graph"], runs["persistent"]
    assert re_ == rg, (name, re_, rg)
    assert (rp["hit_step"], rp["arrival_step"]) == (re_["hit_step"], re_["arrival_step"]), (name, rp, re_)
    assert re_["hit_step"] == ref.hit_time_step
    # arrival steps are recorded until the hit ends the scenario (the oracle's per-device details follow each device's own run)
    for k, d in enumerate(details):
        if d["arrival_step"] != -2 and (ref.hit_time_step == -2 or d["arrival_step"] < ref.hit_time_step):
            assert re_["arrival_step"][k] == d["arrival_step"], (name, k, re_, d)
    if re_["hit_step"] == -2:  # states are specified only when no hit ended the run
        assert np.array_equal(se[0], sg[0]) and np.array_equal(se[1], sg[1])
        assert np.all(np.abs(sp[0] - se[0]) <= 1e-9 * np.abs(se[0]).max())
