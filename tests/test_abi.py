"""The C-ABI shared library loads and exports every symbol include/fatezero_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

from fatezero_amd import _native, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fatezero_hip.h")).read()
    return sorted(set(re.findall(r"\b(fz_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_bound_and_exported():
    declared = _declared()
    assert declared, "no fz_* declarations found"
    assert sorted(_native.exported_symbols()) == declared
    lib = build.build_hip()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", out), f"{name} is not exported by {lib}"


def test_hip_library_loads_or_explains():
    """dlopen of the gfx950 build needs libamdhip64 (present in this image even without a GPU)."""
    lib = build.build_hip()
    h = ctypes.CDLL(lib)
    h.fz_version.restype = ctypes.c_char_p
    assert b"hip gfx950" in h.fz_version()


def test_product_path_fails_loudly_without_the_library(tmp_path, monkeypatch):
    monkeypatch.setattr(_native, "HIP_LIB", str(tmp_path / "libfatezero_hip.so"))
    _native.reset_backend()
    with pytest.raises(_native.NativeLibraryError):
        _native.lib()
    _native.reset_backend()


def test_cpu_tensors_are_refused_by_the_product_backend():
    import torch
    from fatezero_amd import kernels as K
    _native.reset_backend()
    x = torch.zeros(4, 64, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        K.layernorm(x, torch.ones(64, dtype=torch.float16), torch.zeros(64, dtype=torch.float16))


def test_descriptor_layouts_match_the_header():
    # sizes computed by the C compiler for the structs in the header vs the ctypes mirrors
    import tempfile
    prog = r'''
#include <stdio.h>
#include "fatezero_hip.h"
int main(){ printf("%zu %zu %zu\n", sizeof(FzAttnSelfDesc), sizeof(FzAttnCrossDesc), sizeof(FzGemmDesc)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        a, b, c = subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert int(a) == ctypes.sizeof(_native.FzAttnSelfDesc)
    assert int(b) == ctypes.sizeof(_native.FzAttnCrossDesc)
    assert int(c) == ctypes.sizeof(_native.FzGemmDesc)


def test_integration_md_stub_matches_the_compiled_header():
    """INTEGRATION.md shows the ctypes struct a maintainer of the reference would copy.  Extract that very block from the markdown,
    exec it, and compare field names, field offsets and sizeof with what the C compiler makes of include/fatezero_hip.h -- a short
    or re-ordered stub would make the kernel read garbage strides."""
    import tempfile
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"(class FzAttnSelfDesc\(C\.Structure\):.*?\n)\n", md, re.S)
    assert m, "INTEGRATION.md no longer carries the FzAttnSelfDesc stub"
    ns = {"C": ctypes}
    exec(m.group(1), ns)
    stub = ns["FzAttnSelfDesc"]
    names = [f[0] for f in stub._fields_]
    assert names == [f[0] for f in _native.FzAttnSelfDesc._fields_]
    lines = "\n".join(f'    printf("%zu ", offsetof(FzAttnSelfDesc, {n}));' for n in names)
    prog = ('#include <stdio.h>\n#include <stddef.h>\n#include "fatezero_hip.h"\nint main(){\n' + lines +
            '\n    printf("%zu\\n", sizeof(FzAttnSelfDesc)); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        vals = [int(v) for v in subprocess.run([exe], stdout=subprocess.PIPE, text=True, check=True).stdout.split()]
    assert [getattr(stub, n).offset for n in names] == vals[:-1]
    assert ctypes.sizeof(stub) == vals[-1] == ctypes.sizeof(_native.FzAttnSelfDesc)
