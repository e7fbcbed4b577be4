"""N>1 host logic on CPU: two gloo ranks shard clips with no data-path collective and gather ragged wavs."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from tacotronv2_wavernn_chinese_amd.sharding import generate_sharded, shard_indices
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mels = [np.full((80, 21 + i), float(i), np.float32) for i in range(5)]
    calls = []

    def fake_generate(i, mel):  # stands in for model.generate: deterministic, ragged length like (T-1)*275
        calls.append(i)
        return np.arange((mel.shape[1] - 1) * 3, dtype=np.float64) + 1000.0 * i

    out = generate_sharded(fake_generate, mels)
    assert calls == shard_indices(5, rank, world)
    q.put((rank, calls, [o.tolist() for o in out]))
    dist.destroy_process_group()


def test_two_rank_sharding_gathers_every_clip_in_order():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, outs in res:  # every rank holds every wav, in input order, untouched
        assert len(outs) == 5
        for i, o in enumerate(outs):
            np.testing.assert_array_equal(np.asarray(o), np.arange((21 + i - 1) * 3, dtype=np.float64) + 1000.0 * i)


def _worker_ragged(rank, world, port, q):
    import torch.distributed as dist
    from tacotronv2_wavernn_chinese_amd.sharding import balanced_shards, generate_sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lens = [100, 21, 22, 23, 24, 25, 60, 40]
    mels = [np.full((80, t), float(i), np.float32) for i, t in enumerate(lens)]
    seen = []

    def fake_many(idx, ms):   # stands in for model.generate_many: one ragged device call for the rank's whole share
        seen.append(list(idx))
        return [np.arange((m.shape[1] - 1) * 2, dtype=np.float64) + 1000.0 * i for i, m in zip(idx, ms)]

    out = generate_sharded(None, mels, balance=True, generate_many=fake_many)
    assert seen == [balanced_shards(lens, world)[rank]]
    q.put((rank, seen[0], [o.tolist() for o in out]))
    dist.destroy_process_group()


def test_two_rank_length_balanced_sharding_of_ragged_clips():
    """Ragged clips: the length-balanced table (LPT) instead of round-robin, one generate_many call per rank, every wav back
    in input order on every rank."""
    from tacotronv2_wavernn_chinese_amd.sharding import balanced_shards, shard_indices
    lens = [100, 21, 22, 23, 24, 25, 60, 40]
    bins = balanced_shards(lens, 2)
    loads = [sum(lens[i] for i in b) for b in bins]
    rr = [sum(lens[i] for i in shard_indices(len(lens), r, 2)) for r in range(2)]
    assert sorted(i for b in bins for i in b) == list(range(len(lens)))
    assert max(loads) < max(rr) and max(loads) - min(loads) <= 25            # 158 / 157 against round-robin's 206 / 109
    assert balanced_shards([7] * 6, 3) == [[0, 3], [1, 4], [2, 5]]            # equal lengths: the round-robin assignment
    assert balanced_shards([5, 9], 4) == [[1], [0], [], []]                   # more ranks than clips
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == bins
    for _, _, outs in res:
        assert len(outs) == len(lens)
        for i, o in enumerate(outs):
            np.testing.assert_array_equal(np.asarray(o), np.arange((lens[i] - 1) * 2, dtype=np.float64) + 1000.0 * i)


def test_single_process_path():
    from tacotronv2_wavernn_chinese_amd.sharding import generate_sharded
    out = generate_sharded(lambda i, m: np.array([i, m.sum()]), [np.ones((2, 2)), np.zeros((2, 2))])
    assert [o.tolist() for o in out] == [[0, 4.0], [1, 0.0]]


def test_bench_distributed_scaffolding_dry_run(tmp_path):
    """`python bench.py --gpus 2 --config 3` end to end on CPU (gloo) with a stand-in for the device calls: self-launch of the
    ranks, rendezvous on 127.0.0.1, scatter of the clips from rank 0, gather of the waveforms, barrier + MAX-over-ranks timing,
    one JSON line from rank 0 whose gathered waveforms belong to the clips each rank was sent."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    for cfg in ('3', '1'):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--cpu-dry-run', '--config', cfg, '--steps', '2',
                            '--warmup', '1', '--frames', '21', '--batch', '3' if cfg == '3' else '0'],
                           capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
        d = json.loads(lines[0])
        assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['unit'] == 'ksamples/s'
        assert d['data'].startswith('dry-run')
        if cfg == '3':
            assert d['dry_run_check'] == 'ok'
            assert d['value'] == pytest.approx(2 * 2 * 3 * 21 * 275 / (d['ms_per_step'] * 2 / 1e3) / 1e3, rel=1e-3)
        else:   # the driver's SCALE command (config 1 at N > 1) also carries BASELINE configs[3] at that N
            e3 = d['extra_configs']['3']
            assert e3['n_gpus'] == 2 and e3['dry_run_check'] == 'ok' and 'RCCL' in e3['config']['workload']


def test_bench_scaffolding_at_the_real_rank_count(tmp_path):
    """The driver's SCALE run uses 8 ranks; this is the same self-launch at world 8 on CPU (gloo, dry run): rank -> device
    mapping, the scatter of 8 x 2 clips from rank 0, the gather of every rank's waveforms, the barrier and the MAX reduction
    of the elapsed time at the rank count the 8-GPU node will use."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--cpu-dry-run', '--config', '3', '--steps', '2',
                        '--warmup', '1', '--frames', '21', '--batch', '2'], capture_output=True, text=True, timeout=900, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['dry_run_check'] == 'ok' and d['scaling'] == 'weak'
    assert d['value'] == pytest.approx(8 * 2 * 2 * 21 * 275 / (d['ms_per_step'] * 2 / 1e3) / 1e3, rel=1e-3)


def test_the_scale_leg_cannot_take_the_headline_down(tmp_path):
    """`bench.py --gpus 2` (config 1) attaches configs[3]; if that leg hangs -- its RCCL scatter / gather has never run on 8 real GPUs --
    a watchdog prints the headline line without it and every rank leaves.  Here the watchdog is made to fire at once."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['WRNN_BENCH_SCALE_LEG_TIMEOUT'] = '0.001'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--cpu-dry-run', '--config', '1', '--steps', '2', '--warmup', '1',
                        '--frames', '21'], capture_output=True, text=True, timeout=600, env=env, cwd=tmp_path)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0
    assert 'watchdog' in d['extra_configs']['3']['error']
