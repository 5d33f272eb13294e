"""The split sweep keeps two tiles in flight with loads the compiler does not know to be loads (inline assembly) and a wait counted
by hand (nk_kernels.h: NkTileBuf, fetch, arrived).  That is only sound if NOTHING touches a register set between its loads and
the wait that retires them -- a copy the register allocator slips in would read registers the loads have not filled yet.  The
parity tests would see the garbage; this test looks at the code itself: it compiles EVERY split instantiation the dispatch can
reach for gfx950 (no GPU needed) and walks their assembly."""
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


# Every SPLIT instantiation NK_SWEEP_DISPATCH can reach (nk_engine.hip): tables in global memory (GEOM 2: the default reason for
# a split sweep) and, forced with NK_SPLIT=1, in LDS (GEOM 1); rough facets imply ids; RBF temperatures or not; mode records from
# LDS or from the table.  <GEOM, ROUGH, RBF, PID, SPLIT, LREC, FAST>
SPLIT_INSTANCES = ['%d, %s, %s, %s, true, %s, 0' % (g, r, b, p, l)
                   for g in (2, 1) for (r, p) in (('false', 'false'), ('false', 'true'), ('true', 'true'))
                   for b in ('false', 'true') for l in ('true', 'false')]
_ASM = {}


def _assembly():
    """All of them in ONE translation unit (one hipcc run, about a minute), the flags of nanokappa_amd/csrc/Makefile (CXXFLAGS)."""
    if 'lines' not in _ASM:
        assert os.path.exists(HIPCC), 'hipcc is required here: the check is part of the build (ADVICE r3: fail rather than skip)'
        tmp = tempfile.mkdtemp()
        try:
            src = os.path.join(tmp, 'all.hip')
            with open(src, 'w') as f:
                f.write('#define NK_KERNEL_LINKAGE static\n#include <hip/hip_runtime.h>\n#include <stdint.h>\n#include "nk_kernels.h"\n')
                for inst in SPLIT_INSTANCES:
                    f.write('template __global__ void k_sweep<%s>(NkDev, uint32_t, int, int);\n' % inst)
            out = os.path.join(tmp, 'all.s')
            subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics', '-mllvm', '-disable-machine-licm',
                                   '--cuda-device-only', '-I' + os.path.join(ROOT, 'nanokappa_amd', 'csrc'), '-I' + os.path.join(ROOT, 'include'),
                                   '-S', '-o', out, src], stderr=subprocess.DEVNULL)
            _ASM['lines'] = open(out).read().split('\n')
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return _ASM['lines']


def _mangled(inst):
    g, r, b, p, sp, l, fast = [x.strip() for x in inst.split(',')]
    bit = lambda x: 'Lb1E' if x == 'true' else 'Lb0E'
    return '_Z7k_sweepILi%sE%s%s%s%s%sLi%sELb0EEv5NkDevjii' % (g, bit(r), bit(b), bit(p), bit(sp), bit(l), fast)


@pytest.mark.parametrize('inst', SPLIT_INSTANCES)
def test_no_instruction_touches_a_register_set_in_flight(inst):
    lines = _assembly()
    name = _mangled(inst)
    a = next((i for i, l in enumerate(lines) if l.startswith(name + ':')), None)
    assert a is not None, 'kernel %s not found in the assembly' % name
    b = next(i for i in range(a, len(lines)) if 's_endpgm' in lines[i])
    bad, pend, w0, l1 = scan(lines, a, b, [])
    assert w0 is not None and l1 is not None, 'no hand-counted wait found: is the two-ahead form still built for the split sweep?'
    assert not pend, 'loads still in flight at the end of the kernel'
    # once more round the tile loop, entered with what its last turn left in flight
    _, carried, _, _ = scan(lines, a, l1 + 1, [])
    bad2, _, _, _ = scan(lines, w0, l1 + 1, carried)
    msg = '\n'.join('line %d: %s   touches v%s' % (i - a, t, r) for i, t, r in (bad + bad2)[:10])
    assert not bad and not bad2, 'a register set is touched while its loads are in flight:\n' + msg
