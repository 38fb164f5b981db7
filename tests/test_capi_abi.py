"""The C-ABI library loads and exports every symbol include/nbody_amd.h (the run_step boundary) and include/nbody_amd_ext.h
(raw launches, shared pairs, nb_sharded_*, nb_solve_ex) declare (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols(header="nbody_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", text)))


# what a caller of run_step needs and nothing else ("thin C-ABI"): lifecycle, state, nb_step / nb_accel,
# scenarios, nb_solve, state files
CORE = {"nb_abi_version", "nb_device_count", "nb_config_default", "nb_create", "nb_destroy", "nb_strerror", "nb_last_error",
        "nb_set_state", "nb_get_state", "nb_set_mass", "nb_step", "nb_run_step", "nb_accel", "nb_step_timed", "nb_run_scenario",
        "nb_run_scenarios_batched", "nb_restore_snapshot", "nb_save_state", "nb_load_state", "nb_state_file_info",
        "nb_read_state_file", "nb_write_state_file", "nb_solve"}


def test_header_and_binding_agree(nb):
    core, ext = _declared_symbols(), _declared_symbols("nbody_amd_ext.h")
    assert set(core) == CORE, sorted(set(core) ^ CORE)   # the core header stays the run_step boundary
    assert ext and not set(core) & set(ext)
    for name in ("nb_launch_step_f32", "nb_launch_pair_forces_f32", "nb_sharded_create", "nb_sharded_set_deadline", "nb_solve_ex",
                 "nb_selftest_pair_schedule", "nb_plan_shared_pairs_f32", "nb_context_kernel_name"):
        assert name in ext, name
    declared = sorted(core + ext)
    assert sorted(nb.capi.SYMBOLS) == declared
    L = nb.capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.nb_abi_version() == 5
    text = open(os.path.join(ROOT, "include", "nbody_amd.h")).read()
    assert "nb_launch_f32" not in text and "typedef struct nb_sharded" not in text and "nb_solve_options {" not in text
    # the measurement hooks live in their own header and are NOT exported by the product library
    debug = _declared_symbols("nbody_amd_debug.h")
    assert sorted(nb.capi.DEBUG_SYMBOLS) == debug and len(debug) == 3
    for name in debug:
        assert not hasattr(L, name), f"{name} leaked into the product ABI"
    with pytest.raises(nb.capi.NBodyError, match="libnbody_amd_stamps.so"):
        nb.capi._debug_symbol("nb_enable_step_stamps")


def test_struct_layouts_match_header(nb):
    import ctypes as C
    c = nb.capi
    assert C.sizeof(c.NbConfig) == 6 * 4 + 3 * 8
    assert C.sizeof(c.NbScenario) == 6 * 4 + 16 * 4 + 4 * 4 + 16  # ints, watch[], sync_every/engine/flags/graph_chunk, 2 doubles
    assert C.sizeof(c.NbSolveOptions) == 8 * 4
    assert C.sizeof(c.NbStateHeader) == 8 + 4 * 4 + 3 * 8
    assert C.sizeof(c.NbAnswer) == 24
    assert C.sizeof(c.NbLaunchF32) == 7 * 8 + 4 * 8 + 8 * 4 + 3 * 8
    assert C.sizeof(c.NbShardedRank) == 2 * 4 + 2 * 8 + 4 * 4 + 16 + 36 + 64 + 4  # (+4: padded to a multiple of 8)


def test_no_cpu_fallback(nb):
    """On a machine without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert nb.capi.device_count() == 0
    with pytest.raises(nb.capi.NBodyError) as e:
        nb.capi.Context(8)
    assert e.value.code == nb.capi.NB_ERR_NO_DEVICE
    import numpy as np
    z = np.zeros((3, 4))
    with pytest.raises(nb.capi.NBodyError):
        nb.capi.solve(4, 0, 1, z, z, np.ones(4), np.zeros(4, dtype=np.uint8))


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "nthu_ipc_nbody-simulation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text \
                    and "nbody_oracle" not in text, f


def test_state_file_header_errors(nb, tmp_path):
    """nb_state_file_info needs no GPU: bad path / bad magic are I/O errors, not crashes."""
    with pytest.raises(nb.capi.NBodyError) as e:
        nb.capi.state_file_info(str(tmp_path / "missing.nbst"))
    assert e.value.code == nb.capi.NB_ERR_IO
    bad = tmp_path / "bad.nbst"
    bad.write_bytes(b"NOTASTATEFILE" * 8)
    with pytest.raises(nb.capi.NBodyError):
        nb.capi.state_file_info(str(bad))
    import struct
    good = tmp_path / "hdr.nbst"
    hdr = b"NBODYST1" + struct.pack("<qiiddd", 7, 0, 123, 6.674e-11, 1e-3, 60.0)
    good.write_bytes(hdr + bytes(7 * 57))
    assert nb.capi.state_file_info(str(good)) == (7, 0, 123)
    # a header that announces more bodies than the file holds is refused before anyone sizes a buffer from it
    good.write_bytes(hdr + bytes(7 * 57 - 1))
    with pytest.raises(nb.capi.NBodyError, match="truncated"):
        nb.capi.state_file_info(str(good))
    huge = tmp_path / "huge.nbst"
    huge.write_bytes(b"NBODYST1" + struct.pack("<qiiddd", 1 << 40, 0, 0, 6.674e-11, 1e-3, 60.0) + bytes(1000))
    with pytest.raises(nb.capi.NBodyError, match="truncated"):
        nb.capi.read_state_file(str(huge))
    # context-free failures leave their text in the calling thread's nb_last_error(NULL)
    with pytest.raises(nb.capi.NBodyError, match="not an NBODYST"):
        nb.capi.read_state_file(str(bad))
    swapped = tmp_path / "swapped.nbst"
    swapped.write_bytes(b"NBODYST2" + struct.pack(">IiqiiiIddd", 0x01020304, 0, 7, 0, 1, 2, 0, 6.674e-11, 1e-3, 60.0))
    with pytest.raises(nb.capi.NBodyError, match="byte order"):
        nb.capi.read_state_file(str(swapped))


def test_state_file_round_trip_without_gpu(nb, oracle, tmp_path):
    """nb_write_state_file / nb_read_state_file are host-only: the binary form of testcases/b30.in carries exactly the
    text form's content (n, planet, asteroid, q, v, m, the `device` predicate); version-1 files are still readable;
    bin/nbconv writes the same bytes."""
    import subprocess

    import numpy as np
    from conftest import case_path
    c = nb.capi
    s = oracle.read_input(case_path("b30", "in"))
    path = str(tmp_path / "b30.nbst")
    c.write_state_file(path, s.q, s.v, s.m, s.is_device, planet=s.planet, asteroid=s.asteroid)
    assert os.path.getsize(path) == 64 + s.n * (7 * 8 + 1)
    h, q, v, m, dev = c.read_state_file(path)
    assert (h["n"], h["planet"], h["asteroid"], h["precision"], h["step"]) == (s.n, s.planet, s.asteroid, c.NB_F64, 0)
    assert (h["G"], h["eps"], h["dt"]) == (6.674e-11, 1e-3, 60.0)
    assert np.array_equal(q, s.q) and np.array_equal(v, s.v) and np.array_equal(m, s.m) and np.array_equal(dev, s.is_device)
    conv = str(tmp_path / "conv.nbst")
    subprocess.run([os.path.join(ROOT, "bin", "nbconv"), case_path("b30", "in"), conv], check=True)
    assert open(conv, "rb").read() == open(path, "rb").read()
    # version 1 (round-1 checkpoints): 48-byte header, no planet/asteroid
    import struct
    body = open(path, "rb").read()[64:]
    v1 = tmp_path / "v1.nbst"
    v1.write_bytes(b"NBODYST1" + struct.pack("<qiiddd", s.n, 0, 9, 6.674e-11, 1e-3, 60.0) + body)
    h1, q1, _, m1, _ = c.read_state_file(str(v1))
    assert (h1["n"], h1["step"], h1["planet"], h1["asteroid"]) == (s.n, 9, -1, -1)
    assert np.array_equal(q1, s.q) and np.array_equal(m1, s.m)
    trunc = tmp_path / "trunc.nbst"
    trunc.write_bytes(open(path, "rb").read()[:-5])
    with pytest.raises(c.NBodyError, match="truncated"):
        c.read_state_file(str(trunc))


def test_header_is_plain_c_and_links_from_c(nb, tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and a C program must link and call it (no GPU needed
    for the calls made here)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(r"""
#include <stdio.h>
#include "nbody_amd_ext.h"
int main(void) {
    nb_config cfg;
    nb_scenario scn; nb_scenario_result res; nb_answer ans; nb_launch_f32 l; nb_state_header h; nb_solve_options o;
    nb_sharded_rank rk;
    (void)scn; (void)res; (void)ans; (void)l; (void)o; (void)rk;
    if (sizeof(nb_config) != 48 || sizeof(nb_solve_options) != 32 || sizeof(nb_sharded_rank) != 160) return 7;
    if (nb_sharded_rank_info(0, 0, &rk) != NB_ERR_INVALID) return 8;
    if (nb_read_state_file("/nonexistent/x.nbst", &h, 0, 0, 0, 0, 0, 0, 0, 0, 0) != NB_ERR_IO) return 5;
    if (nb_last_error(0)[0] == 0) return 6;
    if (nb_abi_version() != NB_ABI_VERSION) return 1;
    if (nb_config_default(&cfg) != NB_OK || cfg.dt != 60.0 || cfg.G != 6.674e-11 || cfg.eps != 1e-3) return 2;
    if (nb_config_default(0) != NB_ERR_INVALID) return 3;
    if (nb_workspace_bytes_f32(1000, 0) != 18 * 1000 * 16 || nb_workspace_bytes_f32(1000, 1) != 18 * 1000 * 32) return 4;
    printf("%s|%s\n", nb_strerror(NB_OK), nb_strerror(NB_ERR_NO_DEVICE));
    return 0;
}
""")
    exe = tmp_path / "abi"
    libdir = os.path.dirname(nb.capi.library_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-o", str(exe), str(src), "-L", libdir, "-lnbody_amd", f"-Wl,-rpath,{libdir}"], check=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, (p.returncode, p.stderr)
    assert p.stdout.startswith("ok|no usable HIP device")
    # the core header by itself is a complete C99 translation unit too — what INTEGRATION.md's binding includes
    core = tmp_path / "core.c"
    core.write_text(r"""
#include "nbody_amd.h"
int main(void) {
    nb_config cfg; nb_answer ans; nb_scenario scn; (void)ans; (void)scn;
    if (nb_config_default(&cfg) != NB_OK || cfg.flags != 0 || NB_CFG_ORDERED_PAIRS != 1) return 1;
    return nb_abi_version() == 5 ? 0 : 2;
}
""")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    "-o", str(exe), str(core), "-L", libdir, "-lnbody_amd", f"-Wl,-rpath,{libdir}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    binding = open(os.path.join(ROOT, "oracle", "ref_gpu_binding.cc")).read()
    assert "nbody_amd.h" in binding and "nbody_amd_ext.h" not in binding  # the reference-side binding needs the core only


def test_instrumented_build_exports_the_same_abi(nb):
    """libnbody_amd_stamps.so (make stamps: the per-step kernel records clock stamps) is the same ABI, symbol for symbol,
    plus the three hooks of include/nbody_amd_debug.h; that header compiles as C99 too."""
    import ctypes as C
    import subprocess
    path = nb.capi.stamps_library_path()
    assert os.path.exists(path), "make stamps"
    L = C.CDLL(path)
    for name in list(nb.capi.SYMBOLS) + list(nb.capi.DEBUG_SYMBOLS):
        assert hasattr(L, name), name
    L.nb_abi_version.restype = C.c_int
    assert L.nb_abi_version() == 5
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c",
                    os.path.join(ROOT, "include", "nbody_amd_debug.h")], check=True)


def test_text_input_parses_like_the_reference_reads_it(nb, oracle, tmp_path):
    """Property test of the input format (samples/nbody.cc:22-39: `fin >> double`): random systems — huge, tiny, negative,
    subnormal-adjacent and integer-valued numbers, several whitespace layouts, arbitrary type words — written as text with
    17 significant digits must come out of the product's parser (bin/nbconv -> NBODYST2 -> nb_read_state_file, no GPU
    involved) as exactly the doubles that were written, with the `device` predicate and the header indices intact; and
    the oracle's reader (the reference's operator>> restated) must agree."""
    import subprocess

    import numpy as np
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    finite = st.floats(allow_nan=False, allow_infinity=False, width=64)
    body = st.tuples(*([finite] * 7), st.sampled_from(["device", "planet", "asteroid", "body", "Device", "devices", "x"]))
    layout = st.sampled_from([" ", "  ", "\t", "\n"])

    @settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.lists(body, min_size=1, max_size=12), layout, st.data())
    def check(bodies, sep, data):
        n = len(bodies)
        planet, asteroid = data.draw(st.integers(0, n - 1)), data.draw(st.integers(0, n - 1))
        txt = tmp_path / "in.txt"
        with open(txt, "w") as f:
            f.write(f"{n}{sep}{planet}{sep}{asteroid}\n")
            for b in bodies:
                f.write(sep.join("%.17g" % x for x in b[:7]) + sep + b[7] + "\n")
        st_path = tmp_path / "in.nbst"
        subprocess.run([os.path.join(ROOT, "bin", "nbconv"), str(txt), str(st_path)], check=True)
        h, q, v, m, dev = nb.capi.read_state_file(str(st_path))
        want = np.array([b[:7] for b in bodies], dtype=np.float64)
        assert (h["n"], h["planet"], h["asteroid"], h["step"]) == (n, planet, asteroid, 0)
        got = np.concatenate([q, v, m[None, :]], axis=0).T
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (got, want)   # bit for bit, -0.0 included
        assert list(dev) == [int(b[7] == "device") for b in bodies]
        s = oracle.read_input(str(txt))
        assert np.array_equal(np.concatenate([s.q, s.v, s.m[None, :]], axis=0).T.view(np.uint64), want.view(np.uint64))
        assert list(s.is_device) == list(dev) and (s.planet, s.asteroid) == (planet, asteroid)

    check()


def test_output_format_matches_the_reference_stream_formatting(tmp_path):
    """write_output (samples/nbody.cc:41-49: std::scientific << setprecision(digits10 + 1)): the product's writer, driven by
    the sanitizer build of its I/O code, prints random answers — tiny, huge, negative, zero, three-digit exponents — exactly as
    `%.16e` does, which is what the iostream manipulators produce and what the goldens contain."""
    import random
    import subprocess
    subprocess.run(["make", "-C", ROOT, "asan"], check=True, stdout=subprocess.DEVNULL)
    io = os.path.join(ROOT, "bin", "io_check_asan")
    rng = random.Random(7)
    inp = os.path.join(ROOT, "tests", "golden", "testcases", "b20.in")
    for k in range(24):
        d = rng.choice([0.0, 1.0, -1.5, 1e-300, 9.999999999999999e307, 1.1283183768746125e+07]) if k < 6 else \
            rng.uniform(-1, 1) * 10.0 ** rng.randint(-120, 120)
        cost = rng.choice([0.0, 1e5 + 1e3 * 60 * rng.randint(1, 200001)])
        hit, dev = rng.choice([-2, 0, 5, 199999, 200000]), rng.choice([-1, 0, 7, 1023])
        out = tmp_path / "o.out"
        p = subprocess.run([io, inp, str(out), repr(d), str(hit), str(dev), repr(cost)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        assert out.read_text() == "%.16e\n%d\n%d %.16e\n" % (d, hit, dev, cost)


def test_state_file_is_replaced_atomically(nb, tmp_path):
    """A checkpoint overwrites the previous one (bench.py --checkpoint-every: 956 MB at N = 2^24): the new file is written
    beside it and renamed over it, so a write that fails leaves the earlier checkpoint whole and no debris."""
    import numpy as np
    c = nb.capi
    path = str(tmp_path / "ck.nbst")
    q, v, m = np.arange(12.0).reshape(3, 4), np.ones((3, 4)), np.full(4, 2.0)
    c.write_state_file(path, q, v, m, step=5)
    first = open(path, "rb").read()
    assert not os.path.exists(path + ".tmp")
    os.mkdir(path + ".tmp")  # the place the next write needs is taken: that write fails ...
    with pytest.raises(c.NBodyError) as e:
        c.write_state_file(path, q + 1, v, m, step=6)
    assert e.value.code == c.NB_ERR_IO
    assert open(path, "rb").read() == first  # ... and the previous checkpoint is untouched
    os.rmdir(path + ".tmp")
    c.write_state_file(path, q + 1, v, m, step=6)
    h, q2, _, _, _ = c.read_state_file(path)
    assert h["step"] == 6 and np.array_equal(q2, q + 1) and not os.path.exists(path + ".tmp")


def test_bench_multi_gpu_command_needs_no_launcher():
    """`python3 bench.py --gpus 2` typed as is reaches the native C-ABI host: on a machine without a GPU it fails with
    NB_ERR_NO_DEVICE from nb_sharded_create — not with a request for torch.distributed.run, and not by computing on the CPU."""
    import subprocess
    import sys

    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/test_gpu_bench_hosts.py runs the command for real")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "nb_sharded_create" in p.stderr and "no usable HIP device" in p.stderr
    assert "torch.distributed.run" not in p.stderr and not p.stdout.strip()


def test_pair_schedule_of_the_symmetric_kernel_on_shapes_no_box_here_can_run(nb):
    """K1s evaluates every unordered pair of bodies once: every unordered pair of 4096-body superblocks must be met exactly
    once, in all 32 tile phases, over all GPUs, superblocks and workgroups; no slot region may be written twice; the reducer
    must add exactly the slots that were written.  nb_selftest_pair_schedule replays the schedule on the host with the index
    functions the kernels themselves call (csrc/nbody_kernels.h: sym_chunk_range / sym_piece / sym_piece_slot /
    sym_for_each_slot_of) — needs no GPU, so the 8-GPU shapes of BASELINE configs[2]-[4] are checked here."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    c = nb.capi
    SB = 4096
    for n, cus, ranks, acc64 in [(1 << 20, 256, 1, False), (1 << 20, 256, 2, False), (1 << 20, 256, 4, True),
                                 (1 << 20, 256, 8, False),           # configs[2] at 1, 2, 4, 8 GPUs
                                 (1 << 22, 256, 8, False),           # configs[3]
                                 (1 << 24, 256, 8, True),            # configs[4]
                                 (131072 + 5, 256, 1, True), (196608, 256, 1, False), (1500000, 256, 1, False),
                                 (1 << 23, 256, 1, False), (1 << 24, 256, 1, True),   # one GPU, in batches of superblocks
                                 (67 * SB, 256, 1, False), (1 << 18, 304, 1, False), (15 * SB * 8, 256, 3, False)]:
        c.selftest_pair_schedule(n, cus, ranks, acc64)
    with pytest.raises(c.NBodyError, match="cannot share"):
        c.selftest_pair_schedule((1 << 20) + SB, 256, 8)  # shards are not whole superblocks
    with pytest.raises(c.NBodyError, match="cannot share"):
        c.selftest_pair_schedule(1 << 15, 256, 2)         # too few bodies
    with pytest.raises(c.NBodyError, match="does not apply"):
        c.selftest_pair_schedule(28671, 256, 1)

    @settings(max_examples=60, deadline=None)
    @given(st.integers(7, 700), st.integers(0, SB - 1), st.sampled_from([64, 104, 228, 256, 304]), st.booleans())
    def one_gpu(blocks, ragged, cus, acc64):
        if blocks * SB - ragged >= 28672:  # (below that K1s does not apply and the self-test says so)
            c.selftest_pair_schedule(blocks * SB - ragged, cus, 1, acc64)

    @settings(max_examples=40, deadline=None)
    @given(st.integers(2, 8), st.integers(4, 96), st.sampled_from([64, 256, 304]), st.booleans())
    def several_gpus(ranks, per_rank, cus, acc64):
        n = ranks * per_rank * SB
        if n >= 28672 and n * n >= 1.1e9 * ranks:  # (below: a rank's K1 step beats its K1s share, the ranks do not share the pairs)
            c.selftest_pair_schedule(n, cus, ranks, acc64)
        else:
            with pytest.raises(c.NBodyError, match="cannot share"):
                c.selftest_pair_schedule(n, cus, ranks, acc64)

    @settings(max_examples=60, deadline=None)
    @given(st.integers(7, 1100), st.integers(0, SB - 1), st.sampled_from([104, 256, 304]), st.booleans(), st.floats(0.02, 1.0))
    def one_gpu_within_a_budget(blocks, ragged, cus, acc64, share):
        """A workspace smaller than the fastest shape wants (NB_CFG_WORKSPACE_GIB, a raw launch's workspace, a device short of
        memory): batches of superblocks — every pair once, within the budget, or an honest "does not apply"."""
        n = blocks * SB - ragged
        if n < 28672:
            return
        full = c.workspace_bytes_sym_f32(n, acc64)
        if full <= 0:
            return
        try:
            c.selftest_pair_schedule_within(n, max(1 << 20, int(full * share)), cus, acc64)
        except c.NBodyError as e:
            assert "does not apply" in str(e), e   # too small even for batches of 16 superblocks — never an inconsistent schedule

    one_gpu()
    several_gpus()
    one_gpu_within_a_budget()
    for n, gib in ((1 << 20, 1.0), (1 << 22, 8.0), (1 << 22, 3.0), (1 << 24, 20.0)):
        c.selftest_pair_schedule_within(n, int(gib * 2 ** 30))


def test_fp32_modes_refuse_an_eps_whose_inverse_cube_overflows(nb):
    """K1 / K1s evaluate the self pair and the zero-mass padding as d = 0 times G*m*(eps^2)^-1.5: exactly +0 only while that
    power is finite in fp32 and eps^2 is a normal number.  eps = 1e-15 (eps^2 = 1e-30 -> rinv^3 = inf -> 0 * inf = NaN on every
    body) passed round 4's `eps^2 > 0` guard; the ABI now refuses eps < 1e-12 — argument checks, no GPU needed."""
    import ctypes as C
    c = nb.capi
    for eps, ok in ((1e-15, False), (9e-13, False), (1e-23, False), (0.0, False), (1.1e-12, True), (1e-3, True)):
        for prec in (c.NB_F32, c.NB_F32_ACC64):
            try:
                c.Context(64, prec, 0, eps=eps).close()
                refused = False
            except c.NBodyError as e:
                refused = e.code == c.NB_ERR_INVALID   # (NB_ERR_NO_DEVICE on a CPU box = the argument was accepted)
            assert refused == (not ok), (eps, prec)
        try:
            c.Sharded(4096, [0], c.NB_F32, eps=eps).close()
            refused = False
        except c.NBodyError as e:
            refused = e.code == c.NB_ERR_INVALID
        assert refused == (not ok), eps
    # raw launches carry eps^2 as a float
    a = c._launch_struct(1, 1, 4096, 0, 4096, 1e-30, 1e-2, vel_ptr=1)
    assert c.lib().nb_launch_step_f32(C.byref(a), None) == c.NB_ERR_INVALID and "eps2 >= 1e-24" in c.lib().nb_last_error(None).decode()
    # the fp64 mode takes any eps >= 0 (tiny ones run the kernels that skip the self pair explicitly, like nbody.cc:59)
    try:
        c.Context(64, c.NB_F64, 0, eps=1e-15).close()
    except c.NBodyError as e:
        assert e.code == c.NB_ERR_NO_DEVICE
    # unknown nb_config.flags bits are refused; NB_CFG_ORDERED_PAIRS is known
    cfg = c.NbConfig()
    c.lib().nb_config_default(C.byref(cfg))
    cfg.n, cfg.flags = 64, 2
    h = C.c_void_p()
    assert c.lib().nb_create(C.byref(h), C.byref(cfg)) == c.NB_ERR_INVALID


def test_slices_of_small_whole_systems_follow_the_measured_model(nb):
    """plan_f32 for a whole system below 131072 bodies (host logic; 256 CUs when no device answers): the slice counts the
    co-residency model picks are the measured best of profiles/r05_k1_small_n_model.txt at these sizes, within what ONE launch's
    workspace holds (66 records: up to 64 slices; the documented minimum of 18 records: up to 16); a shard of a larger system
    (n_tgt != n_src) follows the same model."""
    c = nb.capi
    best = {1024: 4, 2048: 8, 4096: 16, 6144: 24, 8192: 32, 10240: 20, 12288: 16, 16384: 32, 20480: 12, 22528: 22, 24576: 32,
            26624: 26, 32768: 32, 36864: 21, 40960: 32}
    for n, js in best.items():
        assert c.plan_f32(n, n, workspace_bytes=66 * n * 16, source_path=2) == (4, js, 256), n  # (2 = K1 also where K1s applies)
        r, j, w = c.plan_f32(n, n, workspace_bytes=18 * n * 16, source_path=2)
        assert (r, w) == (4, 256) and j <= 16 and j == min(js, j), n
        assert c.plan_f32(n, n, workspace_bytes=0)[1] == 1  # no workspace, no slices
    assert c.plan_f32(6144, 6144, workspace_bytes=18 * 6144 * 16)[1] == 12 and c.plan_f32(24576, 24576, workspace_bytes=18 * 24576 * 16)[1] == 10
    # shards of a larger system (one rank of 8): the same model, within what one launch holds (measured per rank: 65536 / 8 0.30 -> 0.15 ms)
    assert c.plan_f32(262144, 32768, workspace_bytes=66 * 32768 * 16) == (4, 64, 256)
    assert c.plan_f32(65536, 8192, workspace_bytes=66 * 8192 * 16) == (4, 64, 256)
    assert c.plan_f32(40960, 20480, workspace_bytes=66 * 20480 * 16) == (4, 23, 256)
    assert c.plan_f32(131072, 16384, workspace_bytes=18 * 16384 * 16) == (4, 16, 256)
    assert c.plan_f32(1 << 20, 1 << 20, workspace_bytes=18 * (1 << 20) * 16, j_split=8) == (8, 8, 512)


def test_planned_slices_always_fit_one_launch_of_the_workspace_given(nb):
    """plan_f32 over random shapes (host logic): whatever the model or the older rules pick, a small system's or shard's slice count is at
    least 1, at most one slice per source tile, and — on the 256-thread SGPR path the model covers — at most what ONE launch's workspace
    holds, so that the step stays one force launch + one reducer; without a workspace there are no slices; a forced count is kept."""
    from hypothesis import given, settings, strategies as st
    c = nb.capi

    @settings(max_examples=300, deadline=None)
    @given(st.integers(1, 131071), st.integers(1, 16), st.sampled_from([0, 17, 18, 34, 50, 66, 200]), st.booleans())
    def check(n_tgt, ranks, records, acc64):
        n_src = n_tgt * ranks
        rec = 32 if acc64 else 16
        tpl, js, wg = c.plan_f32(n_src, n_tgt, acc64, workspace_bytes=records * n_tgt * rec)
        ntiles = -(-n_src // 256)
        assert js >= 1 and js <= max(1, ntiles) and tpl in (2, 4, 8) and wg in (256, 512, 1024)
        if records < 18:
            assert js == 1  # fewer than 16 partial-sum slots: no slices
        elif wg == 256 and n_src <= 8_000_000:
            assert js <= min(records - 2, 64), (n_tgt, ranks, records, js)
        if records >= 18 and ntiles >= 4:
            assert c.plan_f32(n_src, n_tgt, acc64, j_split=3, workspace_bytes=records * n_tgt * rec)[1] == 3

    check()
