"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares,
and the ctypes mirrors generated from the header have the C compiler's struct sizes.  No compute calls."""
import ctypes as C
import os

import pytest

from tacotron2_amd import _lib, build


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    assert len(_lib.DECLARED_SYMBOLS) >= 25
    for name in _lib.DECLARED_SYMBOLS:
        assert hasattr(lib, name), name


def test_struct_layouts_match_compiler(lib):
    for name, st in _lib.S.items():
        assert lib.t2_sizeof(name.encode()) == C.sizeof(st), name
    assert lib.t2_sizeof(b"nope") == -1


def test_bad_arguments_return_codes_not_exceptions(lib):
    g = _lib.make("T2Gemm", M=0, N=0, K=0)
    assert lib.t2_gemm(C.addressof(g), None) == 1           # T2_ERR_ARG, no launch attempted
    assert b"t2_gemm" in lib.t2_last_error()
    with pytest.raises(_lib.T2Error):
        _lib.call("t2_gemm", g, None)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(tmp_path, "nope.so"))
    with pytest.raises(_lib.T2Error):
        _lib.lib()


def test_integration_md_stub_mirrors_the_header():
    """The ctypes stub a maintainer would paste from INTEGRATION.md must list every field of T2AttnStep in the header's order (a
    shorter mirror makes the library read garbage behind it)."""
    import re
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = md[md.index("class T2AttnStep(C.Structure)"):]
    block = block[:block.index("]\n") + 1]
    names = re.findall(r'\("(\w+)",\s*C\.', block)
    assert names == [f[0] for f in _lib._structs["T2AttnStep"]], names
