"""The reference's own interface, served by the GPU library.

Mirrors samples/nbody.cc name for name so code (and tests) written against the reference read the same:
  param.*            nbody.cc:9-20
  read_input         nbody.cc:22-39   (returns the vectors instead of filling references)
  write_output       nbody.cc:41-49
  run_step           nbody.cc:51-89   (same argument order; the vectors are updated in place)
  main(argv)         nbody.cc:91-146  + Problem 3 from hw5.cu:438-530,568-602
Every arithmetic operation happens in libnbody_amd.so on the GPU; there is no Python/NumPy compute path.
"""
import math

import numpy as np

from . import capi


class param:  # noqa: N801  (the reference's namespace name)
    n_steps = 200000
    dt = 60.0
    eps = 1e-3
    G = 6.674e-11
    planet_radius = 1e7
    missile_speed = 1e6

    @staticmethod
    def gravity_device_mass(m0, t):
        return m0 + 0.5 * m0 * abs(math.sin(t / 6000))

    @staticmethod
    def get_missile_cost(t):
        return 1e5 + 1e3 * t


def read_input(filename):
    """-> n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, type  (nbody.cc:22-39)."""
    with open(filename) as f:
        tok = f.read().split()
    n, planet, asteroid = int(tok[0]), int(tok[1]), int(tok[2])
    body = tok[3:3 + 8 * n]
    if len(body) != 8 * n:
        raise ValueError(f"{filename}: truncated input")
    cols = [np.array([float(x) for x in body[k::8]]) for k in range(7)]  # float() is correctly rounded
    types = list(body[7::8])
    return (n, planet, asteroid, *cols, types)


def write_output(filename, min_dist, hit_time_step, gravity_device_id, missile_cost):
    """Three lines, scientific with 16 digits (nbody.cc:41-49)."""
    with open(filename, "w") as f:
        f.write("%.16e\n%d\n%d %.16e\n" % (min_dist, hit_time_step, gravity_device_id, missile_cost))


def _is_device(types):
    return np.array([t == "device" for t in types], dtype=np.uint8)


_contexts = {}  # (n, device) -> Context: the reference's run_step is stateless, its GPU stand-in keeps the allocation


def run_step(step, n, qx, qy, qz, vx, vy, vz, m, type, device=0):  # noqa: A002
    """One step on the GPU, updating the six state vectors in place (nbody.cc:51-89).  The signature is the
    reference's, so the state crosses PCIe both ways on every call (like the minimal binding of INTEGRATION.md §2);
    the context (streams, HBM) is created once per system size and reused by later calls."""
    ctx = _contexts.get((n, device))
    if ctx is None:
        ctx = _contexts[(n, device)] = capi.Context(n, capi.NB_F64, device)
    ok = all(isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous for a in (qx, qy, qz, vx, vy, vz))
    if ok:  # nb_run_step: one call, in place, no upload when the vectors still hold what the last call returned
        ctx.run_step(step, qx, qy, qz, vx, vy, vz, m, _is_device(type))
        return
    ctx.set_state(np.stack([qx, qy, qz]), np.stack([vx, vy, vz]), m, _is_device(type))
    ctx.step(step, 1)
    q, v = ctx.get_state()
    qx[:], qy[:], qz[:] = q
    vx[:], vy[:], vz[:] = v


def release_contexts():
    """Free the contexts run_step keeps."""
    while _contexts:
        _contexts.popitem()[1].close()


def solve_file(in_path, out_path, devices=None):
    n, planet, asteroid, qx, qy, qz, vx, vy, vz, m, types = read_input(in_path)
    ans = capi.solve(n, planet, asteroid, np.stack([qx, qy, qz]), np.stack([vx, vy, vz]), m, _is_device(types),
                     devices)
    write_output(out_path, *ans)
    return ans


def main(argv):
    """`prog <in> <out>` (nbody.cc:91-94)."""
    if len(argv) != 3:
        raise RuntimeError("must supply 2 arguments")
    solve_file(argv[1], argv[2])
    return 0
