"""CPU: the C-ABI library builds, loads and exports every symbol include/recstudio_amd.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def nat():
    from recstudio_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    return _native


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'recstudio_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rsa_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_all_exported(nat):
    lib = nat.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in the header but not exported'
    assert sorted(nat.SIGNATURES) == syms          # the ctypes table covers the header, nothing more


def test_abi_version_and_error_channel(nat):
    lib = nat.lib()
    assert lib.rsa_abi_version() == nat.ABI_VERSION == 2
    # argument validation happens before any HIP call, so it can be exercised without a GPU
    rc = lib.rsa_sample_uniform(None, 10, 1, 5, 0, 0, 256, None)
    assert rc == -1 and b'neg_ids is null' in lib.rsa_last_error()
    rc = lib.rsa_sample_uniform(ctypes.c_void_p(8), 10, 5, 5, 0, 0, 256, None)
    assert rc == -1 and b'empty range' in lib.rsa_last_error()
    a = nat.FusedArgs()
    a.dim, a.n_items, a.n_queries, a.num_neg, a.n_query_rows = 6, 10, 2, 1, 2
    assert lib.rsa_fused_sample_gather_score(ctypes.byref(a), None) == -1
    assert b'multiple of 4' in lib.rsa_last_error()
    with pytest.raises(nat.NativeError):
        nat.check(-1, 'x')


def test_struct_layout_matches_header(nat):
    # natural alignment on LP64: 8-byte pointers/int64, 4-byte int32
    assert ctypes.sizeof(nat.FusedArgs) == 8 * 3 + 8 * 2 + 8 + 8 * 2 + 16 + 16 + 8 + 8 * 3 + 8 * 5 + 8 + 8 + 8 * 4 + 8 + 8 + 8 + 8
    assert nat.FusedArgs.seed.offset == 80 and nat.FusedArgs.table.offset == 104
    assert ctypes.sizeof(nat.BackwardArgs) == 8 * 18
    assert nat.BackwardArgs.query.offset == 24 and nat.BackwardArgs.item_grad.offset == 96


def test_missing_library_fails_loudly(nat, monkeypatch):
    monkeypatch.setattr(nat, 'LIB_PATH', '/nonexistent/librecstudio_amd.so')
    monkeypatch.setattr(nat, '_lib', None)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        nat.lib()


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles (a no-op when up to date), loads the library and
    imports the package."""
    import __graft_entry__ as g
    g.build()
