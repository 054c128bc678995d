"""world_size-2 gloo run (CPU) of the multi-GPU sharding layer: stream sharding + birth-count all_gather + gather
to rank 0 reproduce the single-process result (global track IDs included).  The per-rank tracker is the CPU oracle
here (tests may use it as the checker); on the GPU box the same code path runs with the HIP tracker over RCCL."""
import json
import pytest
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_track_fn(packed, iou_thresholds, max_age, min_hits, score_threshold, id_base):
    from oracle import oracle as O
    st = [-np.inf] * len(iou_thresholds) if score_threshold is None else score_threshold
    out = O.track_streams(packed, max_age, min_hits, st, iou_thresholds, id_base)
    births = out.pop('n_births')
    return out, births


def _worker(rank, world_size, port, golden, result_path):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from waymo_2d_tracking_amd import distributed as D
    from waymo_2d_tracking_amd.tracking import utils as T
    exp = json.load(open(os.path.join(golden, 'sort_g4_expected_a.json')))
    p = exp['params']
    predictions = T.read_data_file(os.path.join(golden, 'sort_g4_input.json'), p['score_threshold'])
    rows, total = D.track_all_sharded(predictions, p['iou_threshold'], p['max_age'], p['min_hits'], track_fn=_oracle_track_fn)
    if rank == 0:
        json.dump({'rows': rows, 'total': total}, open(result_path, 'wt'))
    else:
        assert rows is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_tracking_world2_matches_reference(tmp_path, golden_dir, oracle):
    result = str(tmp_path / 'rows.json')
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, golden_dir, result), nprocs=2, join=True)
    got = json.load(open(result))
    exp = json.load(open(os.path.join(golden_dir, 'sort_g4_expected_a.json')))
    assert got['total'] == exp['n_ids']
    key = lambda t: (t['image_id'], t['category_id'], t['object_id'])
    assert [key(t) for t in got['rows']] == [key(t) for t in exp['tracks']]
    gb = np.array([t['bbox'] + [t['score']] for t in got['rows']])
    eb = np.array([t['bbox'] + [t['score']] for t in exp['tracks']])
    np.testing.assert_allclose(gb, eb, rtol=0, atol=1e-6)


def _oracle_merge_fn(packed, k_inputs, method, iou_thresh, soft_nms_cut):
    from oracle import oracle as O
    from waymo_2d_tracking_amd.detnet.ensemble import METHODS
    out5, counts = O.ensemble_groups(packed['dets5'], packed['group_offsets'], packed['input_sizes'], k_inputs, METHODS[method],
                                     iou_thresh, soft_nms_cut)
    return out5, counts


def _collate_worker(rank, world_size, port, golden, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from waymo_2d_tracking_amd import distributed as D
    from waymo_2d_tracking_amd.detnet import ensemble as E
    from waymo_2d_tracking_amd.detnet.trainer import Predictions
    # (a) ragged columnar gather: rank r contributes 3 + 4 r rows (rank 1 more than rank 0), 1-D and 2-D columns
    n = 3 + 4 * rank
    cols = dict(a=np.arange(n, dtype=np.int64) + 100 * rank, b=np.full((n, 4), rank + 0.5), c=np.arange(n, dtype=np.int32))
    got = D.gather_columns_rank0(cols)
    if rank == 0:
        assert got['a'].tolist() == [0, 1, 2] + [100 + i for i in range(7)]
        assert got['b'].shape == (10, 4) and got['b'][:3].max() == 0.5 and got['b'][3:].min() == 1.5 and got['c'].dtype == np.int32
    else:
        assert got is None
    empty = D.gather_columns_rank0(dict(a=np.zeros(0 if rank == 0 else 2, np.float64)))
    assert rank != 0 or empty['a'].shape == (2,)
    # (b) ensemble: images sharded in contiguous blocks, rows gathered to rank 0 == the single-process result (G2)
    exp = json.load(open(os.path.join(golden, 'ensemble_g2_expected.json')))
    files = [os.path.join(golden, 'ensemble_g2_input%d.json' % i) for i in range(3)]
    subs = [E.submission_columns(json.load(open(f))) for f in files]
    image_ids, category_ids, rows = E.merge_inputs(subs, exp['weights'], exp['min_score'])
    out = E.ensemble_columns(image_ids, category_ids, rows, 3, 'soft_nms', exp['iou_thresh'], exp['soft_nms_cut'], exp['min_score'],
                             merge_fn=_oracle_merge_fn)
    if rank == 0:
        json.dump([{'image_id': image_ids[i], 'category_id': int(c), 'bbox': b, 'score': s} for i, c, b, s in
                   zip(out['image'].tolist(), out['category'].tolist(), out['bbox'].tolist(), out['score'].tolist())],
                  open(os.path.join(out_dir, 'ens.json'), 'wt'))
    # (c) inference collation: each rank fills a store for its contiguous block of images, one gather, rank 0 merges
    g6 = json.load(open(os.path.join(golden, 'export_g6.json')))
    ids = list(g6['images'])
    lo, hi = D.contiguous_split(len(ids), world_size)[rank]
    store = Predictions(g6['classnames'], ids)
    for image_id in ids[lo:hi]:
        store[image_id] = [np.asarray(d, np.float32).reshape(-1, 5) for d in g6['predictions'][image_id]]
    cols, tested = store.shard_columns()
    allc = D.gather_columns_rank0(cols)
    allt = D.gather_columns_rank0(dict(tested=tested.astype(np.uint8)))
    if rank == 0:
        from waymo_2d_tracking_amd.detnet import export as X
        full = Predictions.from_shards(g6['classnames'], ids, [allc], [allt['tested'].reshape(world_size, -1).any(0)])
        r = X.detection_rows(full, {k: (v['width'], v['height']) for k, v in g6['images'].items()})
        json.dump(dict(image=[ids[i] for i in r['image']], category=r['category'].tolist(), bbox=r['bbox'].tolist()),
                  open(os.path.join(out_dir, 'inf.json'), 'wt'))
    dist.barrier()
    dist.destroy_process_group()


def test_collation_world2_columns_ensemble_inference(tmp_path, golden_dir, oracle):
    """§8f-2: tensor collation (counts all_gather + one padded gather, no pickling) under gloo, world size 2, for the three
    CLIs' row sets; the sharded ensemble reproduces the reference-generated G2 rows."""
    from test_oracle_ensemble import assert_json_rows_equal
    port = 31500 + os.getpid() % 2000
    mp.spawn(_collate_worker, args=(2, port, golden_dir, str(tmp_path)), nprocs=2, join=True)
    exp = json.load(open(os.path.join(golden_dir, 'ensemble_g2_expected.json')))
    got = json.load(open(tmp_path / 'ens.json'))
    key = lambda r: (r['image_id'], r['category_id'], -r['score'], r['bbox'])
    assert_json_rows_equal(sorted(got, key=key), sorted(exp['outputs']['soft_nms'], key=key))
    g6 = json.load(open(os.path.join(golden_dir, 'export_g6.json')))
    inf = json.load(open(tmp_path / 'inf.json'))
    assert inf['image'] == [r['image_id'] for r in g6['rows']] and inf['bbox'] == [r['bbox'] for r in g6['rows']]
    assert inf['category'] == [r['category_id'] for r in g6['rows']]


def test_splits():
    from waymo_2d_tracking_amd import distributed as D
    assert D.contiguous_split(10, 3) == [(0, 3), (3, 6), (6, 10)]                 # trainer/data/__init__.py:9-18
    b = D.balanced_stream_split([198] * 10, 4)
    assert b[0][0] == 0 and b[-1][1] == 10 and all(b[i][1] == b[i + 1][0] for i in range(3))
    assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
    assert D.balanced_stream_split([5, 1, 1, 1], 2) == [(0, 1), (1, 4)]
    assert D.balanced_stream_split([3], 4)[-1] == (1, 1) or sum(e - s for s, e in D.balanced_stream_split([3], 4)) == 1


def _row_collator_worker(rank, world_size, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from waymo_2d_tracking_amd import distributed as D
    spec = [('frame', torch.int64, ()), ('local_id', torch.int64, ()), ('bbox', torch.float64, (4,)), ('score', torch.float64, ()),
            ('category', torch.int32, ())]
    col = D.DeviceRowCollator(spec, 11, 3, 'cpu')                # capacity 11 rows: the int32 column needs padding to 8 bytes
    for b in range(3):
        k = 2 + 3 * rank + b                                     # ragged: every (rank, block) another row count
        v = col.columns(b)
        v['frame'][:k] = torch.arange(k) + 100 * rank + 10 * b
        v['bbox'][:k] = rank + 0.25 * b
        v['category'][:k] = 1 + rank
        v['score'][:k] = 0.5
        v['local_id'][:k] = torch.arange(k)
        col.counts[b, 0] = k
        col.counts[b, 1] = 7 * (rank + 1) + b                    # births
        col.exchange(b)
    if rank == 0:
        rep = []
        for b in range(3):
            parts, counts = col.decode(b)
            assert counts.tolist() == [[2 + b, 7 + b], [5 + b, 14 + b]]
            for r in range(2):
                k = 2 + 3 * r + b
                assert parts[r]['frame'].tolist() == [i + 100 * r + 10 * b for i in range(k)]
                assert parts[r]['bbox'].shape == (k, 4) and (parts[r]['bbox'] == r + 0.25 * b).all()
                assert parts[r]['category'].dtype == np.int32 and (parts[r]['category'] == 1 + r).all()
            rep.append(counts.tolist())
        json.dump(rep, open(os.path.join(out_dir, 'ok.json'), 'wt'))
    else:
        assert col.collated is None
    dist.barrier()
    dist.destroy_process_group()


def test_device_row_collator_world2(tmp_path):
    """The per-step exchange of the sharded detect+track bench (birth-count all_gather + ONE block gather, no host staging,
    fixed capacity, valid counts travelling next to the rows) under gloo, world size 2."""
    port = 33500 + os.getpid() % 2000
    mp.spawn(_row_collator_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert len(json.load(open(tmp_path / 'ok.json'))) == 3


def test_launcher_environment_and_refusal(tmp_path):
    """bench.py --gpus N: the parent starts N children itself (reference: detnet/trainer/test.py:227-255)."""
    from waymo_2d_tracking_amd import launcher as L
    envs = L.rank_environments(4, 23456, base_env={'PATH': '/usr/bin', 'WORLD_SIZE': 'junk'})
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3'] and [e['LOCAL_RANK'] for e in envs] == ['0', '1', '2', '3']
    assert all(e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '23456' for e in envs)
    assert all(e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and e['PATH'] == '/usr/bin' for e in envs)
    # N > 1: every rank gets its own MIOpen user database / cache and TunableOp result file (8 ranks in find mode on a fresh box
    # must not write one sqlite file); a location the user exported wins; one rank keeps the library defaults
    envs4 = L.rank_environments(4, 23456, base_env={}, scratch=str(tmp_path / 'cache'))
    dirs = [(e['MIOPEN_USER_DB_PATH'], e['MIOPEN_CUSTOM_CACHE_DIR'], e['WT_TUNABLEOP_OUT']) for e in envs4]
    assert len({d[0] for d in dirs}) == 4 and len({d[2] for d in dirs}) == 4
    assert not os.path.exists(tmp_path / 'cache')                  # building the environments has no side effects ...
    for e in envs4:
        L.prepare_rank_caches(e, scratch=str(tmp_path / 'cache'))  # ... the launcher creates the locations, private to this user
    assert all(os.path.isdir(d[0]) and os.path.isdir(d[1]) for d in dirs)
    assert os.stat(tmp_path / 'cache').st_mode & 0o077 == 0 and os.stat(tmp_path / 'cache').st_uid == os.getuid()
    assert L.prepare_rank_caches({'MIOPEN_USER_DB_PATH': str(tmp_path / 'users' / 'db')}, scratch=str(tmp_path / 'cache')) == 0
    assert not os.path.exists(tmp_path / 'users')                  # a location outside the launcher's root is never touched
    kept = L.rank_environments(2, 23456, base_env={'MIOPEN_USER_DB_PATH': '/x', 'OMP_NUM_THREADS': '3'}, scratch=str(tmp_path / 'cache'))
    assert all(e['MIOPEN_USER_DB_PATH'] == '/x' and e['OMP_NUM_THREADS'] == '3' for e in kept)
    assert 'MIOPEN_USER_DB_PATH' not in L.rank_environments(1, 23456, base_env={})[0]
    # a symbolic link planted under the predictable root name is refused (lstat, not stat): its target may be any directory this user owns
    target = tmp_path / 'victim_dir'
    target.mkdir()
    link = tmp_path / 'planted_root'
    os.symlink(target, link)
    with pytest.raises(L.LaunchError):
        L.private_dir(str(link))
    # ranks started by torchrun (the driver's launch) adopt the same per-rank locations in-process; an exported one is left alone
    env = {'MIOPEN_CUSTOM_CACHE_DIR': '/mine'}
    assert sorted(L.adopt_rank_caches(3, 8, environ=env, scratch=str(tmp_path / 'tr'))) == ['MIOPEN_USER_DB_PATH', 'WT_TUNABLEOP_OUT']
    assert env['MIOPEN_CUSTOM_CACHE_DIR'] == '/mine' and 'rank3_of_8' in env['MIOPEN_USER_DB_PATH'] and os.path.isdir(env['MIOPEN_USER_DB_PATH'])
    assert os.path.isdir(os.path.dirname(env['WT_TUNABLEOP_OUT']))
    # a rank's private MIOpen locations start from what a single-process run left in the default ones (same find results on every rank);
    # a destination that already holds files is left alone, a fresh box has nothing to copy
    home = tmp_path / 'home'
    (home / '.config' / 'miopen').mkdir(parents=True)
    (home / '.config' / 'miopen' / 'gfx950.udb.txt').write_text('find results')
    dest = {'MIOPEN_USER_DB_PATH': str(tmp_path / 'seed' / 'db'), 'MIOPEN_CUSTOM_CACHE_DIR': str(tmp_path / 'seed' / 'cache')}
    assert L.seed_rank_cache(dest, home=str(home)) == 1
    assert open(os.path.join(dest['MIOPEN_USER_DB_PATH'], 'gfx950.udb.txt')).read() == 'find results'
    (home / '.config' / 'miopen' / 'gfx950.udb.txt').write_text('newer')
    assert L.seed_rank_cache(dest, home=str(home)) == 0 and open(os.path.join(dest['MIOPEN_USER_DB_PATH'], 'gfx950.udb.txt')).read() == 'find results'
    # the launchers refresh a location THEY seeded when the source has changed since (re-tuned default database)
    os.utime(home / '.config' / 'miopen' / 'gfx950.udb.txt', (2e9, 2e9))
    assert L.seed_rank_cache(dest, home=str(home), refresh=True) == 1
    assert open(os.path.join(dest['MIOPEN_USER_DB_PATH'], 'gfx950.udb.txt')).read() == 'newer'
    assert L.seed_rank_cache(dest, home=str(home), refresh=True) == 0
    assert L.seed_rank_cache({'MIOPEN_USER_DB_PATH': str(tmp_path / 'x')}, home=str(tmp_path / 'nohome')) == 0
    # WT_FORCE_DIST=1 without a launcher: only the rendezvous variables are adopted, a user's thread count survives
    saved = dict(os.environ)
    try:
        os.environ['OMP_NUM_THREADS'] = '5'
        for k in L.RENDEZVOUS_KEYS:
            os.environ.pop(k, None)
        L.adopt_single_rank_env(port=23999)
        assert os.environ['OMP_NUM_THREADS'] == '5' and os.environ['RANK'] == '0' and os.environ['WORLD_SIZE'] == '1'
        assert os.environ['MASTER_PORT'] == '23999' and os.environ['MASTER_ADDR'] == '127.0.0.1'
    finally:
        os.environ.clear()
        os.environ.update(saved)
    with pytest.raises(L.LaunchError, match='only 1 GPU'):
        L.spawn_local_ranks([sys.executable, '-c', 'pass'], 2, n_devices=1)      # never oversubscribe a GPU
    with pytest.raises(L.LaunchError):
        L.spawn_local_ranks([sys.executable, '-c', 'pass'], 0, n_devices=8)
    # children really run, each with its own rank; all succeed -> 0
    code = ("import os; open(os.path.join(%r, 'r' + os.environ['RANK']), 'wt').write(os.environ['WORLD_SIZE'] + ' ' + "
            "os.environ['LOCAL_RANK'] + ' ' + os.environ['MASTER_PORT'])" % str(tmp_path))
    assert L.spawn_local_ranks([sys.executable, '-c', code], 3, n_devices=3) == 0
    got = [open(tmp_path / ('r%d' % r)).read().split() for r in range(3)]
    assert [g[:2] for g in got] == [['3', '0'], ['3', '1'], ['3', '2']] and len({g[2] for g in got}) == 1
    # one failing rank -> its exit code, the sleeping ranks are stopped (not waited for)
    code = "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(600)"
    import time
    t0 = time.time()
    assert L.spawn_local_ranks([sys.executable, '-c', code], 2, n_devices=2) == 7
    assert time.time() - t0 < 60


def test_bench_gpus_flag_refuses_without_gpus():
    """`python bench.py --gpus 8` with fewer GPUs exits non-zero BEFORE any GPU call and never prints a JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8'], env=env, capture_output=True, text=True,
                       timeout=300)
    if torch.cuda.device_count() >= 8:
        return
    assert p.returncode != 0 and 'GPU(s) visible' in (p.stderr + p.stdout) and '"metric"' not in p.stdout
