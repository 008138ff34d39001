"""CPU: libsr_hip.so loads and exports every symbol include/sr_hip.h declares; argument
validation that needs no GPU behaves like the reference's asserts."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from scaling_retriever_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sr_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_status_codes_and_error_strings_without_gpu():
    from scaling_retriever_amd import _lib
    lib = _lib.load()
    assert lib.sr_version() >= 1 and lib.sr_max_topk() >= 1000
    h = ctypes.c_void_p()
    rc = lib.sr_dense_index_create(ctypes.byref(h), 30)
    assert rc == _lib.SR_ERR_INVALID and b"multiple of 16" in lib.sr_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "sr_dense_index_create")
    assert lib.sr_dense_index_create(ctypes.byref(h), 64) == 0
    assert lib.sr_dense_index_ntotal(h) == 0
    assert lib.sr_dense_index_destroy(h) == 0


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from scaling_retriever_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SrHipError):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "scaling_retriever_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "liboracle" not in txt, f
