"""C-ABI checks that need no GPU: the library builds/loads, exports every symbol the header declares, the
header is plain C, and the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
HEADER = os.path.join(ROOT, 'include', 'nanokappa_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nk_[a-z0-9_]+)\s*\(', src)))


def test_header_is_plain_c():
    subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', HEADER])


def test_library_exports_every_declared_symbol():
    from nanokappa_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = engine.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), 'library does not export %s' % n
    assert sorted(engine.EXPORTS) == names


def test_no_cpu_fallback():
    """Without a HIP device nk_create must fail loudly; with one this test is skipped."""
    from nanokappa_amd import engine
    if os.path.exists('/dev/kfd'):
        pytest.skip('GPU present')
    with pytest.raises(engine.NkError) as e:
        engine.Engine(0, 0)
    assert 'no HIP device' in str(e.value) or 'nk_create failed' in str(e.value)


def test_product_does_not_import_oracle():
    """nanokappa_amd/ must never reference oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, 'nanokappa_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'nk_oracle' not in txt and 'nko_' not in txt, f
