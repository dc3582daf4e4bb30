"""The C-ABI library loads and exports every symbol include/onssen_hip.h declares (no compute calls: runs
without a GPU).  Skipped when the library has not been built (hipcc absent)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "onssen_hip.h")).read()
    return sorted(set(re.findall(r"\b(onssen_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from onssen_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(g.OUT):
        pytest.skip("no hipcc and no prebuilt library")
    g.build()
    from onssen_amd.hip import get_lib
    lib = get_lib()
    for name in declared_symbols():
        assert hasattr(lib.dll, name), name
    assert lib.dll.onssen_abi_version() == 14
    assert lib.lstm_geometry(600, 8) == (600, 2400, 38, 75 * 38 * 2 * 256)
    assert lib.dll.onssen_lstm_geometry(600, 7, None, None, None, None) == -1
    assert b"invalid argument" in lib.dll.onssen_error_string(-1)


def test_emu_build_exports_the_same_abi():
    from tests.emu_build import load_emu
    lib = load_emu()
    for name in declared_symbols():
        assert hasattr(lib.dll, name), name


def test_c_host_program_builds_against_the_header_and_the_library():
    """examples/separate_dc.c (C99, no Python, no torch) compiles against include/onssen_hip.h and links to the library: every
    entry point it calls exists with a C-callable signature.  What it computes is checked on the GPU
    (tests/test_gpu_c_abi_example.py)."""
    import subprocess
    import __graft_entry__ as g
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(g.OUT):
        pytest.skip("no hipcc and no prebuilt library")
    g.build()
    exe = os.path.join(ROOT, "examples", "separate_dc")
    assert os.path.exists(exe)
    src = open(os.path.join(ROOT, "examples", "separate_dc.c")).read()
    called = sorted(set(re.findall(r"\b(onssen_[a-z0-9_]+)\s*\(", src)))
    assert len(called) >= 14 and set(called) <= set(declared_symbols())
    undefined = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    for name in called:
        assert name in undefined, name            # resolved from libonssen_hip.so at load time
    # without a GPU it stops at its first HIP call with a message, not a crash
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 1 and "usage" in run.stderr
