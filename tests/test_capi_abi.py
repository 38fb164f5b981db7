"""The C-ABI library loads and exports every symbol include/nbody_amd.h declares (no compute without a GPU)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(nb):
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    assert sorted(nb.capi.SYMBOLS) == declared
    L = nb.capi.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.nb_abi_version() == 1


def test_struct_layouts_match_header(nb):
    import ctypes as C
    c = nb.capi
    assert C.sizeof(c.NbConfig) == 40
    assert C.sizeof(c.NbScenario) == 6 * 4 + 16 * 4 + 4 + 4 + 16  # ints, watch[], sync_every, engine, 2 doubles
    assert C.sizeof(c.NbAnswer) == 24
    assert C.sizeof(c.NbLaunchF32) == 7 * 8 + 4 * 8 + 8 * 4


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
    good.write_bytes(b"NBODYST1" + struct.pack("<qiiddd", 7, 0, 123, 6.674e-11, 1e-3, 60.0))
    assert nb.capi.state_file_info(str(good)) == (7, 0, 123)
