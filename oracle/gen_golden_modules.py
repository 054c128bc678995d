"""TEST INFRASTRUCTURE - generates tests/golden/x152_modules.json from the module tree the reference printed when it
built its Cascade R-CNN X-152-32x8d-FPN dconv model (/root/reference/logs/12442/job.log:336-1221, `print(model)` of
detnet/trainer/train.py).  detectron2 itself is not available, so this printed tree is the only artefact of the reference
that pins the detector's STRUCTURE: every parameter / buffer name (detectron2 state_dict naming, prefixed with `model.` like
the reference's Detectron2Det.model attribute) and shape.

    python oracle/gen_golden_modules.py            # needs /root/reference (this container only)

The output is data (names + shapes), not reference source.
"""
import json
import os
import re
import sys

LOG = '/root/reference/logs/12442/job.log'
FIRST, LAST = 336, 1221
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'x152_modules.json')


def parse_tree(lines):
    """-> list of (path, type, argument string) for every module line `(name): Type(args...`."""
    stack = []      # (indent, name)
    mods = []
    for raw in lines:
        m = re.match(r'^(\s*)\((\w+)\): (\w+)\((.*)$', raw.rstrip('\n'))
        if not m:
            continue
        indent, name, typ, rest = len(m.group(1)), m.group(2), m.group(3), m.group(4)
        while stack and stack[-1][0] >= indent:
            stack.pop()
        stack.append((indent, name))
        mods.append(('.'.join(n for _, n in stack), typ, rest))
    return mods


def entries_of(mods):
    """Parameters / buffers each printed module owns (PyTorch / detectron2 0.1.3 conventions)."""
    out = []
    for path, typ, rest in mods:
        if typ in ('Conv2d', 'DeformConv'):
            if typ == 'Conv2d':
                m = re.match(r'\s*(\d+), (\d+), kernel_size=\((\d+), (\d+)\)', rest)
                if not m:            # multi-line form: "Conv2d(" then "3, 64, kernel_size=..." on the next line
                    continue
                cin, cout, kh, kw = map(int, m.groups())
            else:
                continue
            g = re.search(r'groups=(\d+)', rest)
            groups = int(g.group(1)) if g else 1
            out.append((path + '.weight', [cout, cin // groups, kh, kw], 'param'))
            if 'bias=False' not in rest:
                out.append((path + '.bias', [cout], 'param'))
        elif typ == 'FrozenBatchNorm2d':
            n = int(re.search(r'num_features=(\d+)', rest).group(1))
            for b in ('weight', 'bias', 'running_mean', 'running_var'):
                out.append((path + '.' + b, [n], 'buffer'))
        elif typ == 'GroupNorm':
            n = int(re.match(r'\s*\d+, (\d+)', rest).group(1))
            out.append((path + '.weight', [n], 'param'))
            out.append((path + '.bias', [n], 'param'))
        elif typ == 'Linear':
            fin = int(re.search(r'in_features=(\d+)', rest).group(1))
            fout = int(re.search(r'out_features=(\d+)', rest).group(1))
            out.append((path + '.weight', [fout, fin], 'param'))
            if 'bias=True' in rest:
                out.append((path + '.bias', [fout], 'param'))
    return out


def main():
    if not os.path.exists(LOG):
        sys.exit('needs %s' % LOG)
    lines = open(LOG, errors='replace').read().split('\n')[FIRST - 1:LAST]
    # the printed tree breaks `Conv2d(` / `DeformConv(` with a norm child over several lines: join the argument line
    joined = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if re.search(r'\): (Conv2d|DeformConv)\($', ln.rstrip()):
            joined.append(ln.rstrip() + lines[i + 1].strip())
            i += 2
            continue
        joined.append(ln)
        i += 1
    mods = parse_tree(joined)
    ents = []
    for path, typ, rest in mods:
        if typ == 'DeformConv':
            cin = int(re.search(r'in_channels=(\d+)', rest).group(1))
            cout = int(re.search(r'out_channels=(\d+)', rest).group(1))
            k = int(re.search(r'kernel_size=\((\d+)', rest).group(1))
            g = int(re.search(r'groups=(\d+)', rest).group(1))
            dg = int(re.search(r'deformable_groups=(\d+)', rest).group(1))
            assert dg == 1 and 'bias=False' in rest
            ents.append((path + '.weight', [cout, cin // g, k, k], 'param'))
    ents += entries_of(mods)
    # the printed root is `Detectron2Det( (model): GeneralizedRCNN(` -> paths already start with model.
    ents = [(n, s, k) for n, s, k in ents if n.startswith('model.')]
    order = {n: i for i, (n, _, _) in enumerate(ents)}
    ents.sort(key=lambda e: order[e[0]])
    frozen = ('model.backbone.bottom_up.stem.', 'model.backbone.bottom_up.res2.')      # cfg.MODEL.BACKBONE.FREEZE_AT = 2
    n_param = sum(1 for _, _, k in ents if k == 'param')
    numel = lambda s: int(__import__('functools').reduce(lambda a, b: a * b, s, 1))
    trainable = sum(numel(s) for n, s, k in ents if k == 'param' and not n.startswith(frozen))
    total = sum(numel(s) for n, s, k in ents if k == 'param')
    types = {}
    for _, typ, _ in mods:
        types[typ] = types.get(typ, 0) + 1
    doc = dict(source='logs/12442/job.log:%d-%d (print(model) of the reference run 12442)' % (FIRST, LAST),
               module_type_counts=types, n_entries=len(ents), n_param_tensors=n_param, param_elements=total,
               trainable_elements_freeze_at_2=trainable,
               entries=[[n, s, k[0]] for n, s, k in sorted(ents)])          # [name, shape, 'p'aram | 'b'uffer]
    json.dump(doc, open(OUT, 'w'), separators=(',', ':'))
    print('wrote', OUT, len(ents), 'entries;', types)


if __name__ == '__main__':
    main()
