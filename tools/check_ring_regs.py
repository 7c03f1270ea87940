#!/usr/bin/env python3
"""Static check of the asm-load kernels (conv_pw.h) on their gfx950 disassembly (tools/kernel_meta.sh writes it): the
registers a `global_load_dwordx4` targets ("ring" registers, tied "+v" asm operands) may only ever be read by the
conversion / epilogue arithmetic and LDS stores, and only be written by loads (and the zero-initialisation): a compiler
copy of such a register between its load and the counted s_waitcnt would read stale data.  A kernel passes if that holds
for the whole kernel, or -- where ring registers are only reserved for part of it -- if the first instruction touching every
load's destination is such a consumer behind a vmcnt wait.
    python tools/check_ring_regs.py /tmp/iss_meta/cnn_pw.s [kernel-name substring]"""
import re
import sys
from collections import Counter


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def kernels(path):
    name, body = None, []
    for l in open(path):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', l.strip())
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name:
            body.append(l)
    if name:
        yield name, body


def main():
    path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'pws')
    expect = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # at least this many kernels must match the filter
    ok = True
    checked = 0
    for name, body in kernels(path):
        if filt not in name:
            continue
        checked += 1
        ins, addr, target = [], [], []                # mnemonic + operands, byte offset in the kernel, branch target offset (or None)
        base = None
        for l in body:
            code, _, cmt = l.partition('//')
            code = code.strip()
            if not code:
                continue
            parts = code.replace(',', ' ').split()
            ma = re.match(r'\s*([0-9A-Fa-f]+):', cmt)
            a0 = int(ma.group(1), 16) if ma else None
            if base is None and a0 is not None:
                base = a0
            mt = re.search(r'<[^>]*\+0x([0-9a-fA-F]+)>', cmt) if parts[0].startswith(('s_branch', 's_cbranch')) else None
            ins.append((parts[0], parts[1:]))
            addr.append(a0 - base if a0 is not None and base is not None else None)
            target.append(int(mt.group(1), 16) if mt else None)
        # offsets below which the code runs once (no branch further down jumps back to or above them): copies of a ring register
        # there, BEFORE the first load into it, move its zero-initialisation around -- not data in flight
        back = [t for a, t in zip(addr, target) if t is not None and a is not None and t <= a]
        once_below = min(back) if back else (1 << 62)
        ring = set()
        for op, a in ins:
            if op == 'global_load_dwordx4' and len(a) >= 3 and a[2].startswith('s['):      # the asm loads: SGPR base + 32-bit VGPR offset
                ring |= regs(a[0])
        readers, writers = Counter(), Counter()
        loaded = set()                                # ring registers an asm load has targeted so far (in program order)
        init_copies = 0
        for (op, a), off in zip(ins, addr):
            if not a:
                continue
            dst, src = regs(a[0]), set()
            for x in a[1:]:
                src |= regs(x)
            if op.startswith(('global_store', 'ds_write')):
                src |= dst
                dst = set()
            if op == 'global_load_dwordx4' and len(a) >= 3 and a[2].startswith('s['):
                loaded |= regs(a[0])
            if src & ring and op.startswith('v_mov_b') and not (src & ring & loaded) and off is not None and off < once_below:
                init_copies += 1                      # the zero value of a not-yet-loaded ring register, in the run-once prologue
            elif src & ring:
                readers[op] += 1
            if dst & ring and not (op == 'global_load_dwordx4' and a[2].startswith('s[')):
                writers[op] += 1
        # Second, per-load view (the rule for kernels whose ring registers are NOT reserved for the whole kernel, e.g. the r rows
        # of conv_x3_pwc_kernel that only live across a tile boundary -- their registers serve as temporaries in between, which
        # the whole-kernel view above cannot tell from a copy): the FIRST instruction that touches a load's destination, in
        # program order and around the main loop, must be one of the consumers, behind a vmcnt wait.
        CONSUMERS = ('v_cvt_pk_bf16_f32', 'v_sub_f32', 'v_pk_add_f32', 'v_add_f32', 'ds_write_b128')
        head = 0
        if back:                                      # index of the main loop's head (earliest backward-branch target)
            head = next((i for i, a0 in enumerate(addr) if a0 is not None and a0 >= once_below), 0)
        first = Counter()
        for i, (op, a) in enumerate(ins):
            if not (op == 'global_load_dwordx4' and len(a) >= 3 and a[2].startswith('s[')):
                continue
            R = regs(a[0])
            if not R:
                first['load into AGPRs'] += 1
                continue
            waited, verdict1 = False, 'never read'
            for j in list(range(i + 1, len(ins))) + list(range(head, i)):
                o, b = ins[j]
                if not b:
                    continue
                if o == 's_waitcnt' and any('vmcnt' in x for x in b):
                    waited = True
                src = set()
                for x in b[1:]:
                    src |= regs(x)
                d = regs(b[0])
                if o.startswith(('global_store', 'ds_write')):
                    src |= d
                    d = set()
                if src & R:
                    verdict1 = 'consumed' if (o.startswith(CONSUMERS) and waited) else f"{o}{'' if waited else ' BEFORE A WAIT'}"
                    break
                if d & R:
                    verdict1 = f'overwritten by {o}'
                    break
            first[verdict1] += 1
        per_load_ok = set(first) <= {'consumed'}
        bad_r = {k: v for k, v in readers.items() if not k.startswith(CONSUMERS)}
        bad_w = {k: v for k, v in writers.items() if not k.startswith('v_mov_b')}
        n_init = sum(writers.values())
        scratch = sum(1 for op, a in ins if op.startswith(('scratch_', 'buffer_')))
        whole_ok = not bad_r and not bad_w and n_init * 2 <= len(ring) + 1
        verdict = 'ok' if (whole_ok or per_load_ok) and not scratch else 'CHECK'
        ok &= verdict == 'ok'
        show = (lambda d: dict(d) if len(d) <= 8 else {**dict(list(d.items())[:8]), '...': len(d)})
        print(f"{verdict:5s} {name[:70]:70s} ring regs {len(ring):3d} readers {show(readers)} writers {show(writers)} init copies {init_copies} scratch {scratch} | per load: {dict(first)}")
    if checked < expect:                              # an empty / renamed disassembly must not pass vacuously
        print(f"CHECK {path}: {checked} kernel(s) match {filt!r}, expected at least {expect}")
        ok = False
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
