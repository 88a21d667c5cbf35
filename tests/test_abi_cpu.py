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
    assert h.sfron_abi_version() == built.ABI_VERSION          # _lib.lib() refuses a library of another version
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


def test_descriptor_structs_match_the_header(built):
    """ctypes mirrors of the header's structs: compile a one-line C program against include/sfron.h that prints sizeof / offsetof and
    compare (a silent mismatch would hand the library garbage descriptors)."""
    import ctypes
    import tempfile
    fields = {"sfron_wprep_item": (built.WprepItem, ["w", "fwd", "dgr", "co", "tile0"]),
              "sfron_conv_desc": (built.ConvDesc, ["batch", "n_out", "out_f32", "split_ws"]),
              "sfron_bgemm_desc": (built.BGemmDesc, ["A", "M", "c_f32", "alpha", "split_ws"]),
              "sfron_gemm_desc": (built.GemmDesc, ["A", "M", "a_rowsum", "rowsum_ws", "col_partials"])}
    lines = []
    for cname, (_, fs) in fields.items():
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for f in fs:
            lines.append(f'printf(" %zu", offsetof({cname}, {f}));')
        lines.append('printf("\\n");')
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "sfron.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for line in out:
        name, size, *offs = line.split()
        cls, fs = fields[name]
        assert ctypes.sizeof(cls) == int(size), (name, ctypes.sizeof(cls), size)
        for f, o in zip(fs, offs):
            assert getattr(cls, f).offset == int(o), (name, f, getattr(cls, f).offset, o)


def test_integration_md_stub_prototypes_match_the_header(built, monkeypatch):
    """INTEGRATION.md section 3's ctypes stub (the text a maintainer of the reference would paste): it must load the library, and every
    prototype it declares must have the header's parameter count with pointers where the header has pointers (the round-4 text had six
    pointers where include/sfron.h has seven: moments landed in each other's slots, status 0).  Executed on the GPU by
    tests/test_gpu_sweep_loss.py::test_integration_md_ctypes_stub_runs_as_written."""
    import ctypes
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. Raw C ABI from Python"):text.index("## 4. Entry point")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1
    monkeypatch.chdir(ROOT)
    ns = {}
    exec(compile(blocks[0], "INTEGRATION.md#3", "exec"), ns)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sfron.h")).read(), flags=re.S)
    checked = 0
    for name in ("sfron_sumsq_masked", "sfron_clip_coef", "sfron_masked_clip_adam"):
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", hdr, flags=re.S)
        assert m, name
        params = [p.strip() for p in m.group(1).split(",")]
        fn = getattr(ns["L"], name)
        assert fn.argtypes is not None and len(fn.argtypes) == len(params), (name, len(fn.argtypes or []), len(params))
        for at, p in zip(fn.argtypes, params):
            is_ptr_c = "*" in p
            is_ptr_py = at is ctypes.c_void_p or isinstance(at, type(ctypes.POINTER(ctypes.c_int)))
            assert is_ptr_c == is_ptr_py, (name, p, at)
            if not is_ptr_c:
                want = {"int64_t": ctypes.c_int64, "double": ctypes.c_double, "float": ctypes.c_float, "int": ctypes.c_int}[p.split()[0]]
                assert at is want, (name, p, at)
        checked += 1
    assert checked == 3 and callable(ns["fused_stage"])
