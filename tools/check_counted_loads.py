"""Development: the gathering wavefronts of fc_backward_roles_data_kernel count their own loads (inline asm; hipcc believes the
destination registers are written when the request is made).  This prints every instruction of the kernel's ISA that reads one
of those destination registers, with the instruction before it: each must be the kernel's own copy (v_mov_b32 right behind an
s_waitcnt vmcnt) -- anything else (a phi copy, a spill) would read a row that has not landed.
Usage: python tools/check_counted_loads.py   (compiles fc_backward_ring.hip to ISA and checks every instantiation; CPU only;
tests/test_host_logic.py runs the same check)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEST_RE = re.compile(r'global_load_dwordx2 v\[(\d+):(\d+)\], v\d+, s\[')
NAME_RE = re.compile(r'^_ZN2fc29fc_backward_roles_data_kernelILi(\d+)ELi(\d+)E')


def compile_isa(out='/tmp/fc_backward_ring.s'):
    src = os.path.join(ROOT, 'fieldconv_amd', 'csrc', 'fc_backward_ring.hip')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize', '-w',
                           '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', src, '-o', out])
    return out


def kernels(path):
    """{(R, B): [ISA lines]} for every instantiation of the role-split kernel in the file"""
    found, cur = {}, None
    for ln in open(path):
        m = NAME_RE.match(ln)
        if m and ': ' in ln and '@' in ln:          # the label line `name: ; @name`
            cur = (int(m.group(1)), int(m.group(2)))
            found[cur] = []
        if cur is not None:
            found[cur].append(ln.rstrip())
            if 's_endpgm' in ln:
                cur = None
    return found


def vregs(text):
    used = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', text):
        used.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', text):
        used.add(int(m.group(1)))
    return used


def check(lines, verbose=True):
    """-> (number of destination pairs, list of suspicious reads)"""
    dests = set()
    for l in lines:
        m = DEST_RE.match(l.strip())
        if m:
            dests.add((int(m.group(1)), int(m.group(2))))
    regs = {r for d in dests for r in d}
    load_lines = [i for i, raw in enumerate(lines) if DEST_RE.match(raw.strip())]
    if not load_lines:
        return 0, ['no counted loads found']
    lo, hi = load_lines[0], load_lines[-1] + 40      # the gathering wavefronts' code, from their first request to the last tile's
    # Per basic block: a destination register that the block has not written itself (or has just requested a row into) may hold
    # a row in flight; the only instruction allowed to read it is the kernel's own copy right behind its wait.
    bad, written = [], set()
    for i, raw in enumerate(lines):
        if i < lo or i > hi:
            continue
        l = raw.strip()
        if not raw.startswith('\t') or l.startswith(';') or l.startswith('.'):
            if raw.startswith('.LBB') or raw.startswith('_Z'):
                written = set()
            continue
        op, _, rest = l.partition(' ')
        ops = [o.strip() for o in rest.split(',')]
        store = op.startswith(('global_store', 'scratch_store', 'ds_write', 'buffer_store', 'ds_max', 'ds_add'))
        srcs = ops if store else ops[1:]
        dst = set() if store else vregs(ops[0]) if ops else set()
        used = set()
        for o in srcs:
            used |= vregs(o)
        hit = (used & regs) - written
        if hit and op != 'v_mov_b32':            # (inline-asm spelling of the kernel's own copy: hipcc itself writes v_mov_b32_e32)
            bad.append(f'line {i}: {l}    <- reads v{sorted(hit)} (live into the block or requested in it)')
        if DEST_RE.match(l):
            written -= dst                       # a row is in flight into these
        else:
            written |= dst
        if op.startswith(('s_cbranch', 's_branch', 's_barrier')):
            written = set()
    return len(dests), bad


if __name__ == '__main__':
    found = kernels(compile_isa())
    rc = 0
    for (R, B), lines in sorted(found.items()):
        if B > 2:
            continue                              # (the plan does not take band limit 3: fc_backward_roles.hpp, br_roles_shape_ok)
        n, bad = check(lines)
        print(f'R={R} B={B}: {n} destination pairs, {len(bad)} suspicious reads')
        for b in bad[:10]:
            print('   ', b)
        rc |= 1 if bad or n != 16 else 0
    sys.exit(rc)
