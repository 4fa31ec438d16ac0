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
    assert lib.rsa_abi_version() == nat.ABI_VERSION == 8
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
    structs = (('rsa_fused_args', nat.FusedArgs), ('rsa_backward_args', nat.BackwardArgs),
               ('rsa_shard_route_args', nat.ShardRouteArgs), ('rsa_shard_home_args', nat.ShardHomeArgs))
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
