#!/usr/bin/env python3
"""Static check of the asm-load kernels (conv_pw.h) on their gfx950 disassembly (tools/kernel_meta.sh writes it): the
registers a `global_load_dwordx4` targets ("ring" registers, tied "+v" asm operands) may only ever be read by the
conversion / epilogue arithmetic and LDS stores, and only be written by loads (and the zero-initialisation): a compiler
copy of such a register between its load and the counted s_waitcnt would read stale data.
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
    ok = True
    for name, body in kernels(path):
        if filt not in name:
            continue
        ins = []
        for l in body:
            l = l.split('//')[0].strip()
            if not l:
                continue
            parts = l.replace(',', ' ').split()
            ins.append((parts[0], parts[1:]))
        ring = set()
        for op, a in ins:
            if op == 'global_load_dwordx4' and len(a) >= 3 and a[2].startswith('s['):      # the asm loads: SGPR base + 32-bit VGPR offset
                ring |= regs(a[0])
        readers, writers = Counter(), Counter()
        for op, a in ins:
            if not a:
                continue
            dst, src = regs(a[0]), set()
            for x in a[1:]:
                src |= regs(x)
            if op.startswith(('global_store', 'ds_write')):
                src |= dst
                dst = set()
            if src & ring:
                readers[op] += 1
            if dst & ring and not (op == 'global_load_dwordx4' and a[2].startswith('s[')):
                writers[op] += 1
        bad_r = {k: v for k, v in readers.items() if not k.startswith(('v_cvt_pk_bf16_f32', 'v_sub_f32', 'v_pk_add_f32', 'v_add_f32', 'ds_write_b128'))}
        bad_w = {k: v for k, v in writers.items() if not k.startswith('v_mov_b')}
        n_init = sum(writers.values())
        scratch = sum(1 for op, a in ins if op.startswith(('scratch_', 'buffer_')))
        verdict = 'ok' if not bad_r and not bad_w and not scratch and n_init * 2 <= len(ring) + 1 else 'CHECK'
        ok &= verdict == 'ok'
        show = (lambda d: dict(d) if len(d) <= 8 else {**dict(list(d.items())[:8]), '...': len(d)})
        print(f"{verdict:5s} {name[:70]:70s} ring regs {len(ring):3d} readers {show(readers)} writers {show(writers)} scratch {scratch}")
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
