"""The ping-pong schedule of the 8-row batch kernel (loop_batch.hip, `WRNN_BATCH_PP=1`): OPT-IN.

The variant was written at the end of round 2 after the round's GPU budget was spent, so it has not run on hardware yet;
it is not selected by default (the shipped kernels are ISA-identical with and without it in the source) and these tests
only run with `WRNN_TEST_NEXT=1`.  Every loop of the variant adds in the order of the lock-step 8-row loops, so the contract
is bit-equality with the lock-step kernel -- labels AND fed-back samples -- on top of parity with the reference golden.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden_util import load_case

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('WRNN_TEST_NEXT') != '1', reason='written after the round\'s GPU budget was spent: set WRNN_TEST_NEXT=1')]


def _model(sd, mode='RAW', bits=10):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = bits
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.kernel = _cabi.KERNEL_BATCH
    return m


class _env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _both(m, mels, **kw):
    with _env(WRNN_BATCH_ROWS=8, WRNN_BATCH_PP=0):
        a = m.generate_raw(mels, False, 11000, 550, **kw)
        a = (a['labels'].cpu().numpy(), a['samples'].cpu().numpy())
    with _env(WRNN_BATCH_ROWS=8, WRNN_BATCH_PP=1):
        b = m.generate_raw(mels, False, 11000, 550, **kw)
        b = (b['labels'].cpu().numpy(), b['samples'].cpu().numpy())
    return a, b


def test_pingpong_equals_lockstep_on_the_reference_golden():
    """8 rows on ONE team (both row quads in use), the reference's own noise: lock-step == ping-pong == reference labels."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    fx = load_case('raw_peaky_b8_t60')
    m = _model(fx['state_dict'])
    (la, sa), (lb, sb) = _both(m, fx['mels'], noise_mode=_cabi.NOISE_INJECTED, noise1=fx['noise']['expo'])
    np.testing.assert_array_equal(lb, la)
    np.testing.assert_array_equal(sb, sa)
    np.testing.assert_array_equal(lb.T, fx['labels'].astype(np.int32))


@pytest.mark.parametrize('rows', [64, 13])
def test_pingpong_equals_lockstep_philox(rows):
    """Production noise, all 8 teams with 8 rows each (and a ragged 13 = 8 + 5 rows: masked spare slots, two teams)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, mode='RAW', variant='peaky')
    m = _model(sd)
    mels = make_mels(7, rows, 41)
    (la, sa), (lb, sb) = _both(m, mels, noise_mode=_cabi.NOISE_PHILOX, seed=1234)
    np.testing.assert_array_equal(lb, la)
    np.testing.assert_array_equal(sb, sa)


def test_pingpong_equals_lockstep_mol():
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, mode='MOL', variant='default', bits=9)
    m = _model(sd, mode='MOL', bits=9)
    mels = make_mels(9, 16, 30)
    (la, sa), (lb, sb) = _both(m, mels, noise_mode=_cabi.NOISE_PHILOX, seed=77)
    np.testing.assert_array_equal(sb, sa)


@pytest.mark.parametrize('rpb', [2, 8])
def test_a_team_runs_several_batches_back_to_back(rpb):
    """19 rows with 2 (resp. 8) rows per batch: 10 batches over 8 teams (teams 0 and 1 run two batches one after the other:
    state re-initialised, tags keep counting) resp. 3 batches of 8 + 8 + 3 rows on the 8-row kernel.  Greedy sampling, 3 distinct
    mels in rotation: rows with the same mel must agree whatever batch / team / slot ran them, in both schedules."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, mode='RAW', variant='peaky')
    m = _model(sd)
    base = make_mels(77, 3, 21)
    mels = np.stack([base[i % 3] for i in range(19)])
    for pp in ((0, 1) if rpb == 8 else (0,)):
        with _env(WRNN_BATCH_ROWS=rpb, WRNN_BATCH_PP=pp):
            lab = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX)['labels'].cpu().numpy()
        assert lab.shape == (19, 21 * 275)
        for i in range(3, 19):
            np.testing.assert_array_equal(lab[i], lab[i % 3])
        assert not np.array_equal(lab[0], lab[1])
