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


def test_conv_f32x3_host_geometry_queries():
    """Host-only entry points of the split convolutions: plane sizes follow the tile-major layout [column tile][k chunk][3][BN][16]
    (BN = 64 up to 64 columns, else 128; k padded to 16), and the weight gradient serves powers of two from 64 channels with at most 32 taps."""
    from learning_embeddings_amd import _lib
    lib = _lib.lib
    def elems(ncols, kdim):
        bn = 64 if ncols <= 64 else 128
        return ((ncols + bn - 1) // bn) * ((kdim + 15) // 16) * 3 * bn * 16
    for cout, rs, cin in [(64, 49, 4), (64, 9, 64), (256, 1, 64), (128, 9, 128), (2048, 1, 512), (512, 9, 512), (1000, 1, 40)]:
        assert lib.lec_conv_f32x3_planes_elems(cout, rs, cin, 0) == elems(cout, rs * cin)
        assert lib.lec_conv_f32x3_planes_elems(cout, rs, cin, 1) == elems(cin, rs * cout)
    assert lib.lec_conv_f32x3_wgrad_supported(128, 128, 3, 3) == 1 and lib.lec_conv_f32x3_wgrad_supported(2048, 512, 1, 1) == 1
    assert lib.lec_conv_f32x3_wgrad_supported(64, 256, 1, 1) == 1 and lib.lec_conv_f32x3_wgrad_supported(256, 64, 3, 3) == 1
    assert lib.lec_conv_f32x3_wgrad_supported(192, 128, 1, 1) == 0 and lib.lec_conv_f32x3_wgrad_supported(4, 64, 7, 7) == 0 and lib.lec_conv_f32x3_wgrad_supported(32, 64, 1, 1) == 0


def test_every_ctypes_call_site_passes_the_declared_number_of_arguments():
    """ctypes lets a cdecl call pass MORE arguments than `argtypes` names (and a shifted argument list is a segfault on the GPU box, not a
    Python error): every `lib.lec_*(...)` / `_dt('lec_*', ...)(...)` call in the package, the tests, the tools and bench.py is checked against
    the signature table of learning_embeddings_amd/_lib.py."""
    import ast, glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'learning_embeddings_amd', '_lib.py')).read()
    sig = {m.group(1): len([a for a in m.group(3).split(',') if a.strip()]) for m in re.finditer(r"'(lec_\w+)': \((\w+), \[(.*?)\]\)", src)}
    twins = re.search(r"for base in \((.*?)\):\s+sig\[base \+ '_f32'\]", src, re.S).group(1)
    for base in re.findall(r"'(lec_\w+)'", twins):
        sig[base + '_f32'] = sig[base]
    assert len(sig) > 80
    files = [f for pat in ('learning_embeddings_amd/*.py', 'tests/*.py', 'tools/*.py', 'bench.py', '__graft_entry__.py') for f in glob.glob(os.path.join(root, pat))]
    bad, seen = [], 0
    for f in files:
        for node in ast.walk(ast.parse(open(f).read())):
            if not isinstance(node, ast.Call):
                continue
            fn, name = node.func, None
            if isinstance(fn, ast.Attribute) and fn.attr.startswith('lec_'):
                name = fn.attr
            elif (isinstance(fn, ast.Call) and fn.args and isinstance(fn.args[0], ast.Constant) and str(fn.args[0].value).startswith('lec_')
                  and getattr(fn.func, 'id', getattr(fn.func, 'attr', '')) in ('_dt', 'f')):
                name = fn.args[0].value
            if name in sig and not any(isinstance(a, ast.Starred) for a in node.args):
                seen += 1
                if len(node.args) != sig[name]:
                    bad.append('%s:%d %s passes %d arguments, the ABI takes %d' % (os.path.relpath(f, root), node.lineno, name, len(node.args), sig[name]))
    assert seen > 100 and not bad, '\n'.join(bad)
