#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps *.s) for the fp32-MFMA accumulator-input
write-after-read pattern: a v_mfma_f32_{4x4x1,16x16x4,32x32x2}_f32 whose accumulator input
(src2) is a register range other than its destination, followed within `window` issued
instructions by a vector instruction that writes a register of that input range.
usage: tools/mfma_war_scan.py file.s [kernel-name-substring] [window]"""
import re
import sys
from collections import Counter

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
window = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def regs(tok):
    """v[a:b] / vN / a[..] -> set of (file, index)."""
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


text = open(path).read()
funcs = re.split(r'\n(?=_Z\w+:)', text)
total = Counter()
for f in funcs:
    name = f.split(':', 1)[0]
    if not name.startswith('_Z') or want not in name:
        continue
    lines = [l.strip() for l in f.splitlines()]
    ins = [l for l in lines if l and not l.startswith(('.', ';', '_Z')) and not l.endswith(':')]
    hits = []
    n_mfma = n_untied = 0
    for i, l in enumerate(ins):
        m = re.match(r'(v_mfma_f32_(?:4x4x1|16x16x4|32x32x2)\w*f32)\s+(.*)', l)
        if not m:
            continue
        ops = [o.strip() for o in m.group(2).split(',')]
        dst, src2 = regs(ops[0]), regs(ops[3]) if len(ops) > 3 else set()
        n_mfma += 1
        if not src2 or dst == src2:
            continue
        n_untied += 1
        slots = 0
        for j in range(i + 1, min(i + 1 + 12, len(ins))):
            nxt = ins[j]
            op = nxt.split()[0]
            if op == 's_nop':
                slots += int(nxt.split()[1]) + 1
                continue
            slots += 1
            if slots > window:
                break
            if op.startswith('v_mfma'):
                continue  # a later matrix instruction writes its result after this one has read
            if op.startswith('v_') or op.startswith('ds_') or op.startswith('global_load'):
                first = nxt.split(None, 1)[1].split(',')[0].strip() if ' ' in nxt else ''
                if op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                    continue
                if regs(first) & src2 and not op.startswith(('ds_write', 'ds_bpermute')):
                    hits.append((slots, l, nxt))
                    break
    print('%s: %d fp32 MFMAs, %d with src2 != vdst, %d followed within %d slots by a write to '
          'src2' % (name[:70], n_mfma, n_untied, len(hits), window))
    for slots, a, b in hits[:6]:
        print('    +%d  %s\n        %s' % (slots, a, b))
    total['hits'] += len(hits)
print('total', dict(total))
