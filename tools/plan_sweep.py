"""Planner check of the split-operand kernel: time of the planner's pick against every forced (tile height, K slices) pair, one subprocess per configuration
(the knobs WD_SPLIT_MT / WD_SPLIT_SPLITK are read once per process).

    python tools/plan_sweep.py [gemm M N K | conv B C H W N] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ['gemm 2400 2048 2048', 'gemm 2400 2048 1024', 'gemm 1000 1024 12544', 'gemm 9600 256 1024', 'gemm 2400 256 2048', 'gemm 9600 2048 1024',
           'conv 1 256 80 120 256', 'conv 1 256 40 60 256', 'gemm 38400 256 512', 'gemm 38400 1024 512']


def run(shape, mt, sk):
    env = dict(os.environ)
    env.pop('WD_SPLIT_MT', None); env.pop('WD_SPLIT_SPLITK', None)
    if mt:
        env['WD_SPLIT_MT'] = str(mt)
    if sk:
        env['WD_SPLIT_SPLITK'] = str(sk)
    kind, *dims = shape.split()
    tool = 'gemm_split_one.py' if kind == 'gemm' else 'conv_split_one.py'
    args = dims + (['20', '1'] if kind == 'gemm' else [])
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool)] + args, env=env, capture_output=True, text=True, timeout=300).stdout
    for line in out.splitlines():
        if ' us' in line:
            return float(line.split(':')[-1].split('us')[0])
    return float('nan')


def main():
    shapes = sys.argv[1:] or DEFAULT
    for shape in shapes:
        auto = run(shape, None, None)
        rows = []
        for mt in (4, 5, 6):
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                t = run(shape, mt, sk)
                if t == t:
                    rows.append((t, mt, sk))
        rows.sort()
        print('%-24s planner %.1f us; best forced: %s' % (shape, auto, ', '.join('MT=%d sk=%d %.1f' % (m, k, t) for t, m, k in rows[:4])), flush=True)


if __name__ == '__main__':
    main()
