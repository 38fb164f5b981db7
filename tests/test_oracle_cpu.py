"""CPU-only checks of the oracle itself (the checker must be pinned before it is trusted) and of the host logic."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, case_path, read_golden


@pytest.mark.parametrize("case", ["b20", "b30"])
def test_oracle_cli_reproduces_golden_bytes(oracle, case, tmp_path):
    """BASELINE configs[0]: testcases/b20.in through the CPU path, bit-level diff vs b20.out (all three lines)."""
    out = tmp_path / "o.out"
    exe = os.path.join(ROOT, "oracle", "_build", "nbody_oracle")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run([exe, case_path(case, "in"), str(out)], check=True, env=env)
    assert out.read_bytes() == open(case_path(case, "out"), "rb").read()


@pytest.mark.slow
@pytest.mark.skipif(not os.environ.get("NB_SLOW"), reason="minutes of CPU; set NB_SLOW=1 (gated once, see DESIGN.md)")
@pytest.mark.parametrize("case", ["b40", "b50", "b60", "b70", "b80", "b90", "b100", "b200"])
def test_oracle_cli_reproduces_golden_bytes_slow(oracle, case, tmp_path):
    out = tmp_path / "o.out"
    subprocess.run([os.path.join(ROOT, "oracle", "_build", "nbody_oracle"), case_path(case, "in"), str(out)], check=True)
    assert out.read_bytes() == open(case_path(case, "out"), "rb").read()


@pytest.mark.parametrize("case", ["b20", "b200", "b1024"])
def test_oracle_matches_reference_kats_bitwise(oracle, case):
    """Fixtures were produced by the reference's own run_step (tests/golden/make_kats.py): every bit must agree."""
    kat = np.load(os.path.join(GOLDEN, f"kat_{case}.npz"))
    s = oracle.read_input(case_path(case, "in"))
    done = 0
    for st in kat["steps"]:
        oracle.run_steps(s, done + 1, int(st) - done, omp=True)
        done = int(st)
        assert np.array_equal(s.q, kat[f"q_{st}"]) and np.array_equal(s.v, kat[f"v_{st}"]), (case, st)


def test_oracle_vs_live_reference(oracle):
    """When oracle/_ref is built (build container), step the real reference side by side."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref not built here")
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 130):
        s = oracle.System(n)
        s.q[:] = rng.uniform(-1e12, 1e12, (3, n))
        s.v[:] = rng.uniform(-1e4, 1e4, (3, n))
        s.m[:] = rng.uniform(1e20, 1e30, n)
        s.is_device[::3] = 1
        a, b = s.copy(), s.copy()
        oracle.run_steps(a, 3, 50)
        oracle.ref_run_steps(b, 3, 50)
        assert np.array_equal(a.q, b.q) and np.array_equal(a.v, b.v)
    assert oracle.lib().orc_gravity_device_mass(3.5e20, 1234 * 60.0) == oracle.reflib().ref_gravity_device_mass(3.5e20, 1234 * 60.0)


def test_p3_snapshot_equals_from_zero(oracle):
    """The snapshot shortcut (hw5.cu:265-287) is arithmetic-identical to restarting each device from step 0."""
    s = oracle.read_input(case_path("b20", "in"))
    res, details = oracle.problem23(s, omp=True)
    g = read_golden("b20")
    assert (res.hit_time_step, res.gravity_device_id, res.missile_cost) == (g[1], g[2], g[3])
    d = details[0]
    ok, cost, arr = oracle.problem3_from_zero(s, d["device"], omp=True)
    assert (ok, cost, arr) == (d["feasible"], d["cost"], d["arrival_step"])
    # SURVEY Appendix B-4 known answers
    assert [x["arrival_step"] for x in details] == [127647, 128033]


def test_synthetic_generator_is_counter_based(nb):
    syn = nb.synthetic
    q, v, m = syn.bodies(4096)
    q2, v2, m2 = syn.bodies(4096, 1000, 1100)
    assert np.array_equal(q[:, 1000:1100], q2) and np.array_equal(v[:, 1000:1100], v2) and np.array_equal(m[1000:1100], m2)
    assert np.abs(q).max() < 1 and np.abs(v).max() < 1e-3
    gm = syn.G * m * 4096
    assert gm.min() >= 0.5 and gm.max() < 1.5
    pos, vel = syn.body4_f32(4096, 10, 20)
    assert pos.dtype == np.float32 and pos.shape == (10, 4) and np.allclose(pos[:, 3], syn.G * m[10:20])


def test_host_io_roundtrip(nb, tmp_path):
    n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, types = nb.host.read_input(case_path("b20", "in"))
    assert (n, planet, asteroid) == (20, 2, 17) and types[18] == "device" and types[0] == "black_hole"
    assert qx[0] == -1.5808194255286899e+08 and m[0] == 8.3238852770821595e+36
    out = tmp_path / "x.out"
    g = read_golden("b30")
    nb.host.write_output(str(out), g[0], g[1], g[2], g[3])
    assert out.read_text() == g[4]
    assert nb.host.param.get_missile_cost(60.0) == 1e5 + 6e4
    with pytest.raises(RuntimeError, match="must supply 2 arguments"):
        nb.host.main(["prog"])


def test_host_code_under_sanitizers(tmp_path):
    """ASan + UBSan over the host-side code: the product's text I/O on every testcase input, and the oracle on a
    small full run (GPU AddressSanitizer is unavailable on the pool; the reference's equivalent was cuda-memcheck)."""
    from conftest import ALL_CASES
    subprocess.run(["make", "-C", ROOT, "asan"], check=True, stdout=subprocess.DEVNULL)
    io = os.path.join(ROOT, "bin", "io_check_asan")
    for case in ALL_CASES:
        p = subprocess.run([io, case_path(case, "in")], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        n, planet, asteroid, ndev = [int(x) for x in p.stdout.split()[:4]]
        assert n == int(case[1:]) and ndev in (2, 3, 4)
    out = tmp_path / "w.out"
    g = read_golden("b40")
    p = subprocess.run([io, case_path("b40", "in"), str(out), repr(g[0]), str(g[1]), str(g[2]), repr(g[3])],
                       capture_output=True, text=True)
    assert p.returncode == 0 and out.read_text() == g[4], p.stderr
    # truncated input: must fail cleanly, not read out of bounds
    bad = tmp_path / "bad.in"
    bad.write_text("5 0 1\n1 2 3 4 5 6 7 rock\n1 2 3\n")
    assert subprocess.run([io, str(bad)], capture_output=True).returncode == 1
    # the oracle itself on a 4-body system through all three problems
    small = tmp_path / "s.in"
    small.write_text("4 0 1\n0 0 0 0 0 0 6e24 planet\n4e8 6e7 0 -2e3 0 0 1e12 asteroid\n"
                     "1e8 -2e8 0 0 0 0 3e24 device\n-9e8 9e8 0 0 0 0 1e10 device\n")
    so = tmp_path / "s.out"
    p = subprocess.run([os.path.join(ROOT, "oracle", "_build", "nbody_oracle_asan"), str(small), str(so)],
                       capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert p.returncode == 0, p.stderr
    assert so.read_text().split("\n")[1:3] == ["2506", "2 4.0000000000000000e+05"]


def test_row_list_entry_equals_the_row_range_entry_bitwise(oracle):
    """orc_accel_rows_at (a list of target rows in one call, OpenMP over the list: what the spot checks of large systems use)
    runs the very per-row loop of orc_accel_rows: every bit equal, with and without OpenMP, in any order of the list."""
    import numpy as np
    rng = np.random.default_rng(3)
    n = 3000
    q = np.ascontiguousarray(rng.uniform(-1, 1, (3, n)))
    m = rng.uniform(0.5, 1.5, n)
    rows = [n - 1, 0, 17, 17, 1234, n // 2]
    for omp in (False, True):
        a, ab = oracle.accel_rows_at(q, m, 6.674e-11, 1e-3, rows, want_abs=True, omp=omp)
        for k, i in enumerate(rows):
            r, s = oracle.accel_rows(q, m, 6.674e-11, 1e-3, i, i + 1, want_abs=True, omp=False)
            assert np.array_equal(r[:, 0], a[:, k]) and s[0] == ab[k]
    assert np.array_equal(oracle.accel_rows_at(q, m, 6.674e-11, 1e-3, rows), a)
