"""CPU: the C-ABI library builds, loads and exports every symbol include/recstudio_amd.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'recstudio_amd.h')


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
    assert lib.rsa_abi_version() == nat.ABI_VERSION == 11
    assert lib.rsa_scratch_bytes() >= 256 + 4 * 2048
    # argument validation happens before any HIP call, so it can be exercised without a GPU
    rc = lib.rsa_sample_uniform(None, 10, 1, 5, 0, 0, 256, 0, None)
    assert rc == -1 and b'neg_ids is null' in lib.rsa_last_error()
    rc = lib.rsa_sample_uniform(ctypes.c_void_p(8), 10, 5, 5, 0, 0, 256, 0, None)
    assert rc == -1 and b'empty range' in lib.rsa_last_error()
    a = nat.FusedArgs()
    a.dim, a.n_items, a.n_queries, a.num_neg, a.n_query_rows = 6, 10, 2, 1, 2
    assert lib.rsa_fused_sample_gather_score(ctypes.byref(a), None) == -1
    assert b'multiple of 4' in lib.rsa_last_error()
    with pytest.raises(nat.NativeError):
        nat.check(-1, 'x')


def test_struct_layout_matches_header(nat, tmp_path):
    """Field offsets of the ctypes mirrors == offsetof() in the C header, asked of the C compiler itself."""
    import re
    import shutil
    import subprocess
    hdr = open(HEADER).read()
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    fields = {}
    # every struct the header declares, derived from the header itself -- a new argument block without a checked ctypes mirror
    # fails here
    declared = re.findall(r'typedef struct (rsa_\w+) \{', hdr)
    assert sorted(declared) == sorted(nat.STRUCTS) and len(declared) >= 12
    structs = tuple((name, nat.STRUCTS[name]) for name in declared)
    for struct, cls in structs:
        body = hdr[hdr.index(f'typedef struct {struct} {{'):hdr.index(f'}} {struct};')]
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split('{', 1)[1].split(';'):
            for part in decl.split(','):
                m = re.search(r'(\w+)\s*$', part.strip())
                if m and part.strip():
                    names.append(m.group(1))
        fields[struct] = names
        assert names == [f[0] for f in cls._fields_], f'{struct}: field order differs from the header'
        prog.append(f'  printf("{struct} %zu\\n", sizeof({struct}));')
        prog += [f'  printf("{struct}.{n} %zu\\n", offsetof({struct}, {n}));' for n in names]
    prog += ['  return 0;', '}']
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(prog))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-o', str(exe), str(src)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for struct, cls in structs:
        assert ctypes.sizeof(cls) == int(out[struct])
        for n in fields[struct]:
            assert getattr(cls, n).offset == int(out[f'{struct}.{n}']), f'{struct}.{n}'


def test_versioned_argument_blocks(nat):
    """ABI 9: a block whose first field is `size` is read up to min(size, the library's sizeof): an unset size is refused, a
    block from an OLDER header (shorter) is accepted with the missing fields read as 0, every sized block starts with `size`,
    and no entry point takes more than 12 positional arguments any more."""
    lib = nat.lib()
    sized = [c for c in nat.STRUCTS.values() if issubclass(c, nat._Sized)]
    assert len(sized) >= 6 and all(c._fields_[0] == ('size', ctypes.c_int64) for c in sized)
    assert all(c().size == ctypes.sizeof(c) for c in sized)
    a = nat.LossArgs()
    a.size = 0
    assert lib.rsa_pairwise_loss(ctypes.byref(a), None) == -1 and b'size' in lib.rsa_last_error()
    a = nat.RowsUpdateArgs()                      # a caller compiled before `solo`, `workspace`, ... existed
    a.size = nat.RowsUpdateArgs.exp_avg.offset
    a.n_queries, a.num_neg, a.n_items = 4, 64, 100
    assert lib.rsa_sort_step_elements(ctypes.byref(a), None) == -1
    assert b'neg_ids is null' in lib.rsa_last_error()       # got past the size check, validated the (zero-filled) fields
    a = nat.BprSgdArgs()
    a.n_items, a.n_users, a.num_neg, a.dim = 100, 10, 32, 128
    assert lib.rsa_bpr_sgd_prepare(ctypes.byref(a), None) == -1 and b'num_neg' in lib.rsa_last_error()
    assert max(len(args) for _, args in nat.SIGNATURES.values()) <= 12


def test_sorted_workspace_size_is_monotone(nat):
    """the radix sort's counter region was sized from the tile count of `max_total`, which is not monotone in
    the total (the tile size steps up at every whole round of the chip), while callers sort FEWER elements in a workspace
    sized for the most -- 1 040 000 slots in a workspace sized for 1 072 768 used 1016 tiles where 839 were provided.  The
    size now comes from a monotone upper bound of the tile count."""
    lib = nat.lib()
    prev, prev_sh = 0, 0
    for total in range(4096, 3_000_000, 4096):
        b = lib.rsa_scatter_rows_sorted_workspace_bytes(total, 0, 10_000_000)
        assert b >= prev, total
        prev = b
        sh = lib.rsa_shard_backward_workspace_bytes(total // 1026 + 1, 1026, 4096)
        assert sh >= prev_sh, total
        prev_sh = sh
    # the advisor's example: the workspace sized for slots + Q holds the counters of a sort over the slots alone
    tiles_used = -(-1_040_000 // (256 * 4))                  # 1016 tiles of 256 x RDX_ITEMS_MIN elements
    assert lib.rsa_scatter_rows_sorted_workspace_bytes(1_040_000 + 32_768, 0, 10_000_000) - 2 * ((1_072_768 * 8 + 255) // 256 * 256) \
        >= 256 * 4 * tiles_used


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
