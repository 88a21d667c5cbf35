"""CPU: the C-ABI library loads and exports every symbol include/sfron.h declares (no compute)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from sfron import _lib
    return _lib


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sfron.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfron_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    syms = header_symbols()
    assert len(syms) >= 10
    h = built.lib()
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/sfron.h but not exported by libsfron.so"
    assert sorted(built.declared_symbols()) == syms, "ctypes prototypes and include/sfron.h disagree"
    assert h.sfron_abi_version() >= 1
    assert h.sfron_build_arch() == b"gfx950"


def test_code_object_is_gfx950(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", built.LIB_PATH], capture_output=True, text=True)
    # the fat binary embeds the device code object; its target id must be gfx950
    blob = open(built.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_ops_refuse_cpu_tensors(built):
    import torch
    from sfron import sweep
    with pytest.raises(built.SfronError):
        sweep.mask_from_fisher(torch.zeros(4), torch.zeros(4), 1.0)
