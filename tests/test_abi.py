"""The C-ABI library loads and exports every symbol include/lecone.h declares (no compute calls: runs without a GPU)."""
import ctypes, os, re
import pytest
from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'lecone.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(lec_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from learning_embeddings_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), 'liblecone.so does not export %s' % s
    assert set(_lib.EXPORTS) | {'lec_last_error', 'lec_abi_version'} == set(syms)
    assert _lib.lib.lec_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_not_crashed():
    from learning_embeddings_amd import _lib
    rc = _lib.lib.lec_pair_energy_fwd(7, None, 0, None, 0, 4, 10, 0.1, None, None)
    assert rc == _lib.E_ARG and b'unknown energy' in _lib.lib.lec_last_error()
    rc = _lib.lib.lec_table_step_adam(None, None, None, None, 10, 5, 10, 1e-3, 0.9, 0.999, 1e-8, 1, 0.1, 1, 1, None)
    assert rc == _lib.E_ARG
    assert _lib.lib.lec_loss_workspace_bytes(256, 5, 10) >= 256
    assert _lib.lib.lec_loss_workspace_bytes(256, 5, 5000) < 0          # embedding_dim beyond the kernel's range
    with pytest.raises(_lib.LeconeError):
        _lib.check(rc)


def test_cpu_tensor_is_refused():
    import torch
    from learning_embeddings_amd import ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.pair_energy(torch.rand(4, 10), torch.rand(4, 10))
