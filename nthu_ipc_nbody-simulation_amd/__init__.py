"""nbody_amd — MI355X-native direct N-body step, drop-in for dasbd72/NTHU_IPC_Nbody-Simulation's hot path.

The directory is named ``nthu_ipc_nbody-simulation_amd`` (not an importable identifier); ``import nbody_amd``
(the shim module at the repository root) loads it under that name.

Layers:
  csrc/          HIP kernels for gfx950 + the C ABI (include/nbody_amd.h) + the ``hw5`` CLI host
  capi           ctypes binding of libnbody_amd.so — fails loudly when the library or a GPU is missing
  host           the reference's own interface on top of it: param, read_input, write_output, run_step, main
  synthetic      seeded counter-based generator of the benchmark inputs
  distributed    index-sharded multi-GPU stepper: one process per GPU, RCCL all-gather of positions per step
"""
from . import capi, host, synthetic  # noqa: F401  (distributed imports torch: import it explicitly)
from .capi import NBodyError, library_path  # noqa: F401

__all__ = ["capi", "host", "synthetic", "NBodyError", "library_path"]
