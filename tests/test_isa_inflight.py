"""The split sweep keeps two tiles in flight with loads the compiler does not know to be loads (inline assembly) and a wait counted
by hand (nk_kernels.h: NkTileBuf, fetch, arrived).  That is only sound if NOTHING touches a register set between its loads and
the wait that retires them -- a copy the register allocator slips in would read registers the loads have not filled yet.  The
parity tests would see the garbage; this test looks at the code itself: it compiles two instantiations for gfx950 (no GPU
needed) and walks their assembly."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(lines, start, stop, pending):
    """Walks lines[start:stop]; `pending` = register sets of the hand-issued loads still in flight, oldest first.  Returns
    (violations, pending at the end, line of the first hand-written wait, line of the last hand-issued load)."""
    bad, in_asm, first_wait, last_load = [], False, None, None
    for i in range(start, stop):
        l = lines[i]
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            continue
        code = t.split(';')[0]
        if in_asm and code.startswith('global_load_dword'):
            dest = code.split(',')[0]
            pending.append(regs_of(dest))
            last_load = i
            addr = ','.join(code.split(',')[1:])
            hit = regs_of(addr) & set().union(*pending[:-1]) if len(pending) > 1 else set()
            if hit:
                bad.append((i, t, sorted(hit)))
            continue
        m = re.match(r's_waitcnt\s+.*vmcnt\((\d+)\)', code)
        if m:
            n = int(m.group(1))
            if in_asm:
                if first_wait is None and n > 0:
                    first_wait = i
                pending[:] = pending[len(pending) - n:] if n else []
            elif n == 0:
                pending[:] = []                      # the compiler drained the counter: everything has arrived
            continue
        if code.startswith('s_waitcnt'):
            continue
        flight = set().union(*pending) if pending else set()
        hit = regs_of(code) & flight
        if hit:
            bad.append((i, t, sorted(hit)))
    return bad, pending, first_wait, last_load


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
@pytest.mark.parametrize('inst', ['2, true, false, true, true, true, 0', '1, false, false, false, true, true, 0'])
def test_no_instruction_touches_a_register_set_in_flight(inst):
    tmp = tempfile.mkdtemp()
    try:
        src = os.path.join(tmp, 'one.hip')
        with open(src, 'w') as f:
            f.write('#define NK_KERNEL_LINKAGE static\n#include <hip/hip_runtime.h>\n#include <stdint.h>\n#include "nk_kernels.h"\n'
                    'template __global__ void k_sweep<%s>(NkDev, uint32_t, int, int);\n' % inst)
        out = os.path.join(tmp, 'one.s')
        # the flags of nanokappa_amd/csrc/Makefile (CXXFLAGS)
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-mllvm', '-disable-machine-licm',
                               '--cuda-device-only', '-I' + os.path.join(ROOT, 'nanokappa_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'),
                               '-S', '-o', out, src], stderr=subprocess.DEVNULL)
        lines = open(out).read().split('\n')
        a = next(i for i, l in enumerate(lines) if l.startswith('_Z7k_sweep'))
        b = next(i for i in range(a, len(lines)) if 's_endpgm' in lines[i])
        bad, pend, w0, l1 = scan(lines, a, b, [])
        assert w0 is not None and l1 is not None, 'no hand-counted wait found: is the two-ahead form still built for the split sweep?'
        assert not pend, 'loads still in flight at the end of the kernel'
        # once more round the tile loop, entered with what its last turn left in flight
        _, carried, _, _ = scan(lines, a, l1 + 1, [])
        bad2, _, _, _ = scan(lines, w0, l1 + 1, carried)
        msg = '\n'.join('line %d: %s   touches v%s' % (i - a, t, r) for i, t, r in (bad + bad2)[:10])
        assert not bad and not bad2, 'a register set is touched while its loads are in flight:\n' + msg
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
