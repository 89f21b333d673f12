"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
symbol include/omnisafe_amd.h declares; the ctypes table mirrors the header; the product refuses to
run without a GPU instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'omnisafe_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(osa_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from omnisafe_amd import _lib, build

    build.build_library(verbose=False)
    lib = ctypes.CDLL(build.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/omnisafe_amd.h but not exported'
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes table and header disagree'


def test_no_compute_free_calls():
    from omnisafe_amd import _lib

    lib = _lib.load()
    assert lib.osa_version() == 1
    assert lib.osa_build_arch() == b'gfx950'
    assert lib.osa_strerror(-3) == b'not implemented in libomnisafe_amd'
    assert lib.osa_reduce_ws_bytes() > 0
    # argument validation happens before any launch
    assert lib.osa_gae_scan(None, None, None, None, None, None, None, 4, 4, 0.99, 0.95, 0.95, 0.0, 0,
                            None, None, None, None, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_fails_loudly_without_gpu():
    import numpy as np

    from omnisafe_amd import _lib
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from omnisafe_amd.spaces import Box

    with pytest.raises(_lib.OsaError):
        VectorOnPolicyBuffer(Box(-np.inf, np.inf, (3,)), Box(-1, 1, (2,)), 4, 0.99, 0.95, 0.95, 'gae',
                             0.0, True, True, num_envs=2, device='cuda:0')


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'omnisafe_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dp, f)).read()
                assert 'np_oracle' not in txt and 'ref_harness' not in txt, f
