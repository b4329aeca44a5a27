"""CPU-only: the C-ABI shared library builds, loads and exports every symbol the header
declares (no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "hyphy_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hyphy_hip_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from hyphy_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip.EXPORTS) == names


def test_no_device_means_hard_error_not_fallback():
    """Without a GPU the compute entry points must fail loudly (< 0), never compute on the CPU."""
    import numpy as np
    from hyphy_amd import hip
    if hip.device_count() > 0:
        return
    try:
        hip.expm_batch(np.zeros((1, 4, 4)))
    except hip.HipError as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("expm_batch computed something without a device")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "hyphy_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle/hbl.py", "").replace("``oracle/hbl.py``", ""), f
