"""libdg16.so loads (no GPU needed) and exports every symbol include/dg16.h declares; creating a
context without a GPU fails loudly instead of falling back."""

import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dg16.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dg16_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    import dg16_amd
    from dg16_amd.lib import load, EXPORTED
    L = load()
    decl = declared_symbols()
    assert len(decl) >= 15
    for s in decl:
        assert hasattr(L, s), "include/dg16.h declares %s but libdg16.so does not export it" % s
    assert sorted(EXPORTED) == decl, "python binding list and header disagree"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dg16_amd
    with pytest.raises(dg16_amd.Dg16Error):
        dg16_amd.Context(0)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under distributed-groth16_amd/ may import, link or
    dlopen it."""
    pkg = os.path.join(ROOT, "distributed-groth16_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "liboracle", "oracle/c", "corc"):
                    assert needle not in txt, "%s references the oracle (%s)" % (f, needle)
