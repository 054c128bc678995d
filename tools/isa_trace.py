#!/usr/bin/env python
"""Compact instruction-class trace of a kernel from hipcc -S output (tools: which instructions sit between MFMA bursts).
    python tools/isa_trace.py file.s kernel_substring [start_line end_line]"""
import re
import sys


def cls(op):
    if op.startswith('v_mfma'):
        return 'MFMA'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'DSR'
    if op.startswith('ds_'):
        return 'DSW'
    if op.startswith('global_load_lds') or op.startswith('buffer_load') and 'lds' in op:
        return 'DMA'
    if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'):
        return 'VLD'
    if op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('flat_store'):
        return 'VST'
    if op.startswith('global_atomic'):
        return 'VAT'
    if op.startswith('s_waitcnt'):
        return 'WAIT'
    if op.startswith('s_barrier'):
        return 'BAR'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'BR'
    if op.startswith('s_nop'):
        return 'NOP'
    if op.startswith('v_pk_'):
        return 'VPK'
    if op.startswith('v_'):
        return 'V'
    if op.startswith('s_'):
        return 'S'
    return '?'


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l and l.rstrip().endswith(':') or (name in l and l.strip().endswith(': ; @' + l.split(':')[0].strip()) if False else False)) \
        if False else next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + name + r'\w*:', l))
    out, run, cnt, detail = [], None, 0, []
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith('.Lfunc_end'):
            break
        if not l or l.startswith(';') or l.startswith('.') and not l.startswith('.LBB'):
            continue
        if l.startswith('.LBB'):
            if run:
                out.append((run, cnt, detail)); run, cnt, detail = None, 0, []
            out.append(('LABEL ' + l.split(':')[0], 0, []))
            continue
        op = l.split()[0]
        c = cls(op)
        if c == 'WAIT':
            c = 'WAIT[' + ' '.join(l.split()[1:]) + ']'
        if c == run:
            cnt += 1
        else:
            if run:
                out.append((run, cnt, detail))
            run, cnt, detail = c, 1, []
    if run:
        out.append((run, cnt, detail))
    line = []
    for r, c, _ in out:
        if r.startswith('LABEL'):
            if line:
                print(' '.join(line)); line = []
            print(r)
        else:
            line.append('%s%s' % (r, 'x%d' % c if c > 1 else ''))
            if len(line) >= 14:
                print(' '.join(line)); line = []
    if line:
        print(' '.join(line))


if __name__ == '__main__':
    main()
