"""The C-ABI boundary: libase_hip.so loads on a machine without a GPU and exports every function that
include/ase_hip.h declares, with the arity ase_amd/lib.py binds (no compute calls here)."""
import ctypes
import os
import re

import pytest

from ase_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'ase_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|const char\*)\s+(ase_hip_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('void', '') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_is_built_and_loads():
    assert os.path.exists(L.LIB_PATH), "run __graft_entry__.build() first"
    lib = L.load()
    assert lib.ase_hip_abi_version() == 1
    assert lib.ase_hip_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    decls = _header_decls()
    assert len(decls) >= 25
    lib = ctypes.CDLL(L.LIB_PATH)
    for name, nargs in decls.items():
        assert hasattr(lib, name), f'{name} declared in ase_hip.h but not exported'
        if name in ('ase_hip_abi_version', 'ase_hip_last_error'):
            continue
        assert name in L.SIGNATURES, f'{name} has no ctypes binding in ase_amd/lib.py'
        assert len(L.SIGNATURES[name]) == nargs, (name, len(L.SIGNATURES[name]), nargs)
    for name in L.SIGNATURES:
        assert name in decls, f'{name} bound in lib.py but not declared in the header'


def test_argument_validation_fails_loudly_without_gpu():
    """Null operands are rejected by the host-side checks before any launch (works without a GPU)."""
    lib = L.load()
    rc = lib.ase_hip_gemm_nt(None, 0, None, 0, None, 0, None, None, 0, 0, 0, None, 0, None, 0, 0, 0, 0, 0, 0, 0, 1.0, L.BF16, None)
    assert rc == -1 and b'gemm_nt' in lib.ase_hip_last_error()
    with pytest.raises(L.AseHipError):
        L.check(rc, 'gemm_nt')


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ase_amd.backend import HipBackend
    with pytest.raises(L.AseHipError):
        HipBackend()
