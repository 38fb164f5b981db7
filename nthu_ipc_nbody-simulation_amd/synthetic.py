"""Seeded, counter-based synthetic bodies — any rank can generate any slice with no communication.

u(i,k) = (splitmix64(seed + 7*i + k) >> 11) * 2^-53 ; position 2u-1 (k=0..2), velocity (2u-1)*1e-3 (k=3..5),
mass (0.5+u)/(N*G) (k=6) so that G*m is in [0.5,1.5)/N and |a| = O(1).  No `device` bodies.
"""
import numpy as np

SEED = 42
G = 6.674e-11
EPS = 1e-3
DT = 1e-4
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform(i0, i1, k, seed=SEED):
    idx = np.arange(i0, i1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.uint64(7) * idx + np.uint64(k)
        z = _splitmix64(x)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def bodies(n_total, i0=0, i1=None, seed=SEED):
    """-> q (3,cnt), v (3,cnt), m (cnt,) in fp64 for bodies [i0,i1) of an n_total-body system."""
    i1 = n_total if i1 is None else i1
    q = np.stack([2.0 * uniform(i0, i1, k, seed) - 1.0 for k in range(3)])
    v = np.stack([(2.0 * uniform(i0, i1, k, seed) - 1.0) * 1e-3 for k in range(3, 6)])
    m = (0.5 + uniform(i0, i1, 6, seed)) / (n_total * G)
    return q, v, m


def body4_f32(n_total, i0=0, i1=None, seed=SEED):
    """float4 records {x,y,z,G*m} and {vx,vy,vz,0} as (cnt,4) float32 arrays (the fp32 HBM layout)."""
    q, v, m = bodies(n_total, i0, i1, seed)
    pos = np.empty((q.shape[1], 4), dtype=np.float32)
    pos[:, :3] = q.T
    pos[:, 3] = G * m
    vel = np.zeros((q.shape[1], 4), dtype=np.float32)
    vel[:, :3] = v.T
    return pos, vel
