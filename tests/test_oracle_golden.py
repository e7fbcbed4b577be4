"""Pins the C restatement (oracle/wavernn_oracle.c) to golden vectors minted
from the unmodified reference (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import ALL_CASES, load_case


def _oracle_run(fx):
    om = orc.OracleModel(fx['state_dict'], mode=fx['mode'], bits=int(fx['bits']))
    cm, ca = om.conditioning(fx['mels'])
    if fx['batched']:
        cm = om.fold(cm, int(fx['target']), int(fx['overlap']))
        ca = om.fold(ca, int(fx['target']), int(fx['overlap']))
    if fx['mode'] == 'RAW':
        r = om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'])
    else:
        r = om.loop(cm, ca, 0, fx['noise']['u_mix'], fx['noise']['u_log'])
    return om, cm, ca, r


@pytest.mark.parametrize('name', ALL_CASES)
def test_prologue_matches_reference(name):
    fx = load_case(name)
    om = orc.OracleModel(fx['state_dict'], mode=fx['mode'], bits=int(fx['bits']))
    up, aux = om.conditioning(fx['mels'])
    np.testing.assert_allclose(up[:, :320], fx['up_head'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(up[:, -320:], fx['up_tail'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(up[:, ::41], fx['up_stride'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(aux[:, ::275], fx['aux_frames'], rtol=0, atol=5e-6)


@pytest.mark.parametrize('name', ALL_CASES)
def test_loop_and_epilogue_match_reference(name):
    fx = load_case(name)
    om, cm, ca, r = _oracle_run(fx)
    assert cm.shape[:2] == (fx['rows'], fx['L'])
    if fx['mode'] == 'RAW':
        # bit-exact class indices over the whole free-running sequence
        np.testing.assert_array_equal(r['labels'], fx['labels'].astype(np.int32))
        ncls, mu = 1024, True
    else:
        np.testing.assert_allclose(r['samples'], fx['samples'], rtol=0, atol=2e-6)
        ncls, mu = 30, False
    wave_len = (int(fx['T']) - 1) * 275
    wav = orc.epilogue(r['samples'].T, ncls, mu, bool(fx['batched']), int(fx['target']), int(fx['overlap']),
                       wave_len, 275)
    assert wav.shape == fx['wav'].shape and wav.dtype == np.float64
    np.testing.assert_allclose(wav, fx['wav'], rtol=0, atol=1e-6 if fx['mode'] == 'MOL' else 0)


def test_teacher_forcing_reproduces_free_run():
    fx = load_case('raw_peaky_b1_t24')
    om, cm, ca, r = _oracle_run(fx)
    r2 = om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'], x_forced=r['samples'], want_logits=True)
    np.testing.assert_array_equal(r2['labels'], r['labels'])
    assert np.isfinite(r2['logits']).all()


def test_epilogue_rejects_short_clips():
    # the reference crashes for T < 21 (broadcast error at fatchord_version.py:258)
    with pytest.raises(ValueError):
        orc.epilogue(np.zeros((1, 20 * 275), np.float32), 1024, True, False, 11000, 550, 19 * 275, 275)


@pytest.mark.parametrize('name', ['raw_peaky_b1_t401', 'raw_peaky_b8_t60'])
def test_baseline_size_goldens_pin_the_fast_oracle(name):
    """The BASELINE-size fixtures (configs[1]'s 110 275-step clip; B=8, T=60) were minted from the unmodified reference
    (oracle/make_golden.py long): the OpenMP/AVX2 build of the restatement -- the checker of the GPU tests at those
    sizes -- reproduces the reference's labels bit for bit over the whole free-running sequence, and its wav."""
    from tests.golden_util import LONG_CASES
    assert name in LONG_CASES
    fx = load_case(name)
    om = orc.OracleModel(fx['state_dict'], fast=True)
    cm, ca = om.conditioning(fx['mels'])
    r = om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'])
    np.testing.assert_array_equal(r['labels'], fx['labels'].astype(np.int32))
    if fx['wav'].size:
        wav = orc.epilogue(r['samples'].T, 1024, True, False, 11000, 550, (int(fx['T']) - 1) * 275, 275)
        np.testing.assert_array_equal(wav.astype(np.float32), fx['wav'])


def test_torch_cpu_loop_is_the_reference_loop():
    """oracle/torch_cpu_loop.py (bench.py's `cpu_baseline`: the reference's op sequence on torch-CPU, timed on the GPU box where
    /root/reference does not exist) reproduces the labels the unmodified reference produced -- same ops, same RNG consumption."""
    from oracle import torch_cpu_loop as tl
    fx = load_case('raw_peaky_b1_t24')
    r = tl.run(fx['state_dict'], fx['mels'], 1500, 2, seed=int(fx['noise_seed']))
    np.testing.assert_array_equal(r['labels'], fx['labels'].astype(np.int64)[:1500])
