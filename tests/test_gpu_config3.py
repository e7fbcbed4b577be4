"""BASELINE configs[3] (throughput mode) through bench.py's own pipeline on one GPU: rank 0 owns the clips, `dist.scatter`s the
mels, the rank generates its share on the batch kernel, finishes every utterance with ONE `wrnn_epilogue_rows` launch and the
float64 waveforms are `dist.gather`ed back -- over a one-rank RCCL communicator, i.e. the same NCCL calls the 8-GPU run makes
(the first multi-GPU run is then not also the first RCCL run).  The waveforms that come back are checked against the oracle:
labels along the GPU's trajectory (every step), waveform = the oracle's float64 epilogue of those samples.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.parity_util import check_on_gpu_trajectory_raw

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_scatter_generate_epilogue_gather_on_one_rank(tmp_path):
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    from tests.test_gpu_baseline_sizes import _forced_raw, _philox_q
    B, T = 16, 41
    dump = tmp_path / 'c3.npz'
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--config', '3', '--frames', str(T), '--batch', str(B),
                        '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--dump', str(dump)],
                       capture_output=True, text=True, timeout=900, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['config']['kernel'] in ('batch', 'batch_cs') and 'RCCL' in d['config']['workload']
    z = np.load(dump)
    wave, lab, smp = z['wave'], z['labels'], z['samples']
    L, wave_len = T * 275, (T - 1) * 275
    assert wave.shape == (1, B, wave_len) and wave.dtype == np.float64 and lab.shape == (B, L)
    seed = int(z['seed'][0])
    assert seed == 0xC0FFEE                                  # step 0 of rank 0 (bench.py: seed_of)
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1000, B, T)                             # rank 0's slice of the scattered job
    om = orc.OracleModel(sd, fast=True)
    rows = list(range(B))
    st = check_on_gpu_trajectory_raw(lab.T, smp.T, _forced_raw(om, mels, rows, _philox_q(seed, L, rows)))
    from tests.parity_util import bound_near_ties
    margins = [float(st['ref']['margin'][t, r]) for t, r, _ in st['near_ties']]
    bound_near_ties(f'configs[3] scatter/generate/epilogue/gather, world 1 (fp32-oracle margins at the near-ties: {margins})', st['compared'], st['near_ties'])
    assert st['compared'] == B * L
    # the gathered waveform of every utterance = the oracle's float64 tail of generate() (:243-258) on that row's samples
    for b in range(B):
        ref = orc.epilogue(smp[b:b + 1], 1024, True, False, 11000, 550, wave_len, 275)
        np.testing.assert_allclose(wave[0, b], ref, rtol=0, atol=4 * np.finfo(np.float64).eps)
    assert torch.cuda.is_available()
