"""CPU: the C-ABI library builds/loads and exports every symbol include/wcx.h declares; the
product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from wisecondorx_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "wcx.h")).read()
    declared = set(re.findall(r"\b(wcx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"wcx_ctx", "wcx_ref"}
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    L = lib.load()
    for name in declared:
        assert hasattr(L, name)
    assert L.wcx_version() >= 100


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.WcxError, match="no HIP device|no CPU fallback|hip"):
        lib.Context(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under wisecondorx_amd/ may import, load, link
    or execute it."""
    pkg = os.path.join(ROOT, "wisecondorx_amd")
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(\boracle\s*\.)|(libwcx_oracle)|(wcx_oracle)|(oracle/)",
                     re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not bad.search(txt), f
