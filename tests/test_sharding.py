"""N>1 host logic on CPU: two gloo ranks shard clips with no data-path collective and gather ragged wavs."""
import os
import socket

import numpy as np
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


def test_single_process_path():
    from tacotronv2_wavernn_chinese_amd.sharding import generate_sharded
    out = generate_sharded(lambda i, m: np.array([i, m.sum()]), [np.ones((2, 2)), np.zeros((2, 2))])
    assert [o.tolist() for o in out] == [[0, 4.0], [1, 0.0]]
