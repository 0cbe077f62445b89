"""CPU suite: the product library loads and exports every symbol the C ABI header
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from poseestimation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(ph_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 10
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    assert declared_symbols() == set(_lib.SIGNATURES)


def test_calls_fail_loudly_without_init():
    lib = _lib.load()
    if lib.ph_device() >= 0:
        return  # a GPU test in this process already bound the library
    st = lib.ph_stream_synchronize(None)
    assert st == 4  # PH_ERR_NOT_INITIALIZED
    assert b"ph_init" in lib.ph_last_error_string()


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "poseestimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, fn)


def test_only_test_code_loads_the_oracle():
    """Outside tests/ the checker may be loaded by __graft_entry__.smoke() and by bench.py's
    cpu_baseline legs only: no tool, example or header names its binding or its library."""
    for top in ("tools", "examples", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if fn.endswith((".py", ".sh", ".hip", ".h", ".cpp", ".hpp")):
                    text = open(os.path.join(dirpath, fn)).read()
                    for token in ("oracle_bind", "liboracle", "oracle/"):
                        assert token not in text, (os.path.join(dirpath, fn), token)
