"""CPU: the C-ABI shared library loads and exports every symbol include/auxssm.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "auxssm.h")).read()
    return sorted(set(re.findall(r"\b(auxssm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from aux_ssm_samplers_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/auxssm.h but not exported"
    assert set(_lib.exported_symbols()) == set(declared), set(_lib.exported_symbols()) ^ set(declared)
    assert lib.auxssm_version() == 108


def test_no_gpu_fails_loudly():
    """Without a GPU the product must raise, never fall back to a CPU path."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    from aux_ssm_samplers_amd import _lib
    with pytest.raises((_lib.AuxSSMError, ValueError)):
        _lib.Handle(0)


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aux_ssm_samplers_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                s = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+(oracle|tests)\b", s, re.M) or re.search(r"import_module\([\"']oracle", s):
                    bad.append(f)
    assert not bad, bad
