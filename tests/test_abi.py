"""CPU-side checks of the C-ABI boundary: the in-tree library loads and exports every declared symbol.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

from crab_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    out = set()
    for f in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", f)).read()
        out |= set(re.findall(r"^\s*(?:int|void|const char\*)\s+(crab_[a-z0-9_]+)\s*\(", txt, flags=re.M))
    return out


def test_library_built_in_tree():
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations found in include/*.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    missing = declared - set(_lib.SYMBOLS)
    assert not missing, f"ctypes bindings missing for {missing}"
    assert lib.crab_abi_version() >= 1


def test_ops_fail_loudly_without_gpu_tensors():
    import torch
    from crab_amd import ops
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.CrabHipError):
        ops.gemm(x, x)
