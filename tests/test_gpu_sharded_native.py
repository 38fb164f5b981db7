"""nb_sharded_*: the index-sharded stepper behind the C ABI (one process, P GPUs, in-place ncclAllGather per GPU per
step).  A one-GPU box can run P = 1 through exactly that code — RCCL communicator, all-gather call and all — where the
trajectory must equal nb_step's bit for bit; P > 1 is the driver's multi-GPU run (bin/nbody_bench N steps warmup f32 P)."""
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
    assert (info["targets_per_lane"], info["j_split"], info["wg_size"]) == \
        c.plan_f32(n, n, prec == c.NB_F32_ACC64, workspace_bytes=c.workspace_bytes_f32(n, prec == c.NB_F32_ACC64))
    assert np.abs(q1 - q).max() > 1e-6


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
