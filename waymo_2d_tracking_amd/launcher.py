"""One process per GPU, started by the parent itself - the way the reference starts its N inference workers
(/root/reference/detnet/trainer/test.py:227-255: one spawned process per dataset split, `cuda_device_id = i % n_gpu`;
detnet/trainer/launch.sh:3-9 for the training side).

`spawn_local_ranks` is used by `bench.py --gpus N` when no launcher set WORLD_SIZE (the CLIs run under `torchrun`; their
`-j N` is the number of loader threads per rank, like the reference's `--jobs`): the parent
must not have touched the GPU (no HIP call, no `torch.cuda.is_available()`): it only counts devices (which does not
initialise the runtime on this image), starts N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in their environment, forwards their output, and fails if any child fails.  Children are plain
`subprocess` children of a parent that never initialised HIP - no fork of GPU state, no exec from a GPU process.
"""
import os
import socket
import subprocess
import sys
import time


class LaunchError(RuntimeError):
    pass


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Number of GPUs this process could use, without initialising the HIP runtime."""
    import torch
    return int(torch.cuda.device_count())


RENDEZVOUS_KEYS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')


CACHE_KEYS = ('MIOPEN_USER_DB_PATH', 'MIOPEN_CUSTOM_CACHE_DIR', 'WT_TUNABLEOP_OUT')
SEED_MARKER = '.wt_seeded_from'


def rank_cache_root(scratch=None):
    import tempfile
    return scratch or os.path.join(tempfile.gettempdir(), 'wt_rank_cache_%d' % os.getuid())


def per_rank_cache_env(rank, n_ranks, scratch=None):
    """Library caches that N ranks of one node must not share on a fresh box: every rank runs MIOpen's find mode for the same
    convolutions during warm-up and would write the same user database (sqlite, one writer), and TunableOp's result file is
    written at exit.  One directory / file per rank, under `scratch` (default: wt_rank_cache_<uid> in the system temp directory).
    Pure: builds names, creates nothing (prepare_rank_caches does)."""
    d = os.path.join(rank_cache_root(scratch), 'rank%d_of_%d' % (rank, n_ranks))
    return {'MIOPEN_USER_DB_PATH': os.path.join(d, 'miopen'), 'MIOPEN_CUSTOM_CACHE_DIR': os.path.join(d, 'miopen_cache'),
            'WT_TUNABLEOP_OUT': os.path.join(d, 'tunableop.csv')}


def private_dir(path):
    """Create `path` for this user only (mode 0700) and refuse one that somebody else owns or may write: the default root has a
    predictable name in a world-writable directory, and what is read from it ends up selecting kernels."""
    import stat
    os.makedirs(path, mode=0o700, exist_ok=True)
    st = os.lstat(path)                      # lstat: a symlink planted under the predictable name must not pass for the directory it points to
    if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode):
        raise LaunchError('%s is a symbolic link or not a directory: refusing to keep library caches there' % path)
    if st.st_uid != os.getuid():
        raise LaunchError('%s belongs to uid %d, not to this user: refusing to keep library caches there' % (path, st.st_uid))
    if st.st_mode & 0o022:
        os.chmod(path, st.st_mode & 0o7755 & ~0o022)
    return path


def _newest_mtime(root):
    newest = 0.0
    for base, _, files in os.walk(root):
        for f in files:
            try:
                newest = max(newest, os.stat(os.path.join(base, f)).st_mtime)
            except OSError:
                pass
    return newest


def seed_rank_cache(env, home=None, refresh=False):
    """Start a rank's private MIOpen locations from what a single-process run on this box left in the default ones (find results in
    ~/.config/miopen, compiled kernels in ~/.cache/miopen): the ranks then skip the find / compile work of the warm-up AND pick the same
    solvers (find-mode results vary run to run, which would make some ranks slower than others).  Nothing to copy on a fresh box.  A
    destination that already holds files is left alone - unless `refresh` is set (the launchers set it) AND it was seeded by this function
    from a source that has changed since (SEED_MARKER holds the source's newest mtime at that time): after re-tuning the default database
    the ranks follow it instead of keeping stale find results.  To start over by hand: delete the root (rank_cache_root()).
    The source is copied while nobody else should be writing it (sqlite without its -wal file): run the single-process warm-up first.
    Returns the number of directories seeded."""
    import shutil
    home = home or os.path.expanduser('~')
    n = 0
    for key, default in (('MIOPEN_USER_DB_PATH', os.path.join(home, '.config', 'miopen')),
                         ('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(home, '.cache', 'miopen'))):
        dest = env.get(key)
        if not dest or not os.path.isdir(default) or os.path.abspath(dest) == os.path.abspath(default):
            continue
        try:
            stamp = _newest_mtime(default)
            marker = os.path.join(dest, SEED_MARKER)
            if os.path.isdir(dest) and os.listdir(dest):
                seeded = float(open(marker).read()) if refresh and os.path.exists(marker) else None
                if seeded is None or stamp <= seeded:
                    continue
                shutil.rmtree(dest)
            shutil.copytree(default, dest, dirs_exist_ok=True)
            with open(marker, 'wt') as f:
                f.write(repr(stamp))
            n += 1
        except (OSError, ValueError):
            pass                                     # a cache is an optimisation: the rank starts cold instead
    return n


def prepare_rank_caches(env, scratch=None, home=None):
    """The side effects rank_environments / per_rank_cache_env do not have: create the per-rank locations named in `env` (the root private to
    this user) and seed them.  Only locations under the launcher's own root are touched - one the user exported is the user's business."""
    root = os.path.abspath(rank_cache_root(scratch))
    mine = {k: env[k] for k in CACHE_KEYS if k in env and os.path.abspath(env[k]).startswith(root + os.sep)}
    if not mine:
        return 0
    private_dir(root)
    for k, v in mine.items():
        os.makedirs(os.path.dirname(v) if k == 'WT_TUNABLEOP_OUT' else v, mode=0o700, exist_ok=True)
    return seed_rank_cache(mine, home=home or env.get('HOME'), refresh=True)


def adopt_rank_caches(rank, n_ranks, environ=None, scratch=None):
    """In-process form of per_rank_cache_env for ranks somebody else started (torchrun): point this process's library caches at its
    own directories before MIOpen / TunableOp initialise.  A location the user exported wins.  Returns the keys it set."""
    environ = os.environ if environ is None else environ
    taken = []
    for k, v in per_rank_cache_env(rank, n_ranks, scratch).items():
        if k in environ:
            continue
        environ[k] = v
        taken.append(k)
    prepare_rank_caches({k: environ[k] for k in taken}, scratch)
    return taken


def rank_environments(n_ranks, port, base_env=None, scratch=None):
    """The environment of every child: torchrun's variables for a single node, rendezvous on 127.0.0.1 (the
    container's hostname may not resolve); values the user already set (thread counts, IPC mode, cache locations) win.
    Pure: builds dictionaries; spawn_local_ranks creates / seeds the per-rank cache locations (prepare_rank_caches)."""
    base = dict(os.environ if base_env is None else base_env)
    envs = []
    for r in range(n_ranks):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                 MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WT_LAUNCHED_BY='waymo_2d_tracking_amd.launcher')
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC only on this pool (RCCL needs it)
        e.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n_ranks)))
        if n_ranks > 1:
            for k, v in per_rank_cache_env(r, n_ranks, scratch).items():
                e.setdefault(k, v)
        envs.append(e)
    return envs


def adopt_single_rank_env(port=None):
    """WT_FORCE_DIST=1 without a launcher: make this process rank 0 of a one-rank group.  Only the rendezvous variables are written
    into os.environ - a thread count or IPC mode the user exported stays as it is."""
    env = rank_environments(1, port or free_port(), {})[0]
    for k in RENDEZVOUS_KEYS:
        os.environ[k] = env[k]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def spawn_local_ranks(argv, n_ranks, port=None, n_devices=None, poll_s=0.2, timeout_s=None):
    """Start `n_ranks` children running `argv` (e.g. [sys.executable, 'bench.py', '--gpus', '8', ...]), one per GPU.
    Returns 0 when every child exits 0.  Raises LaunchError before starting anything when fewer than n_ranks GPUs are
    visible; returns the first non-zero exit code (after terminating the remaining children) when a child fails."""
    if n_ranks < 1:
        raise LaunchError('--gpus must be >= 1, got %d' % n_ranks)
    have = visible_gpus() if n_devices is None else n_devices
    if have < n_ranks:
        raise LaunchError('%d ranks requested but only %d GPU(s) visible: refusing to oversubscribe a GPU '
                          '(one process per GPU)' % (n_ranks, have))
    port = port or free_port()
    envs = rank_environments(n_ranks, port)
    for e in envs:
        prepare_rank_caches(e)
    procs = [subprocess.Popen(list(argv), env=e) for e in envs]
    t0 = time.time()
    rc = 0
    try:
        live = set(range(n_ranks))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write('launcher: rank %d exited with code %d; stopping the other ranks\n' % (r, code))
            if rc != 0:
                break
            if timeout_s is not None and time.time() - t0 > timeout_s:
                rc = 124
                sys.stderr.write('launcher: timeout after %.0f s\n' % timeout_s)
                break
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:                                          # exact PIDs we started, never a pattern
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
