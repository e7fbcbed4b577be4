"""Parity at the BASELINE.json config sizes, through the C-ABI, on the shipped kernels with default settings.

configs[1]: B=1, mel 80x401 (110 275 free-running steps = 14 default segments of the latency kernel), RAW 10-bit;
configs[2]: B=64 utterances (batch kernel, 8 rows per XCD team in lock-step), sampled with the production Philox noise;
configs[4]: MOL 9-bit, B=32 (batch kernel, 4 rows per team), injected uniforms.
Oracle = the C restatement pinned to the reference (tests/test_oracle_golden.py); the configs[1] clip is additionally
compared with labels minted from the unmodified reference itself (tests/golden/raw_peaky_b1_t401.npz).

The north star's contract is "within +-1 LSB at 10-bit": every test here asserts label IDENTITY up to the first near-tie
of the sampler's race (tests/parity_util.py) and prints max |label difference| over the compared steps, so the log shows
the contract directly (0 = bit-identical fed-back values).
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.golden_util import load_case
from tests.parity_util import MOL_LSB, check_free_run_raw, check_mol, label_stats

pytestmark = pytest.mark.gpu


def _model(sd, mode='RAW', bits=10, kernel='auto'):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = bits
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.kernel = {'auto': _cabi.KERNEL_AUTO, 'team2': _cabi.KERNEL_TEAM2, 'batch': _cabi.KERNEL_BATCH, 'simple': _cabi.KERNEL_SIMPLE}[kernel]
    return m


def _oracle_rows(om, mels, rows, noise_mode, n1=None, n2=None):
    """Oracle on a subset of the batch rows (rows are independent: state is per row, :194-196)."""
    cm, ca = om.conditioning(mels[rows])
    return om.loop(cm, ca, noise_mode, None if n1 is None else np.ascontiguousarray(n1[:, rows]),
                   None if n2 is None else np.ascontiguousarray(n2[:, rows]))


def test_config1_b1_t401_injected_noise_vs_reference_and_oracle():
    """configs[1]: the benchmarked clip, free-running over all 110 275 steps with the reference's own noise stream
    (452 MB of Exp(1) draws on the device), default segmentation (14 launches), against the labels of the unmodified
    reference and against the oracle."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    fx = load_case('raw_peaky_b1_t401')
    m = _model(fx['state_dict'])
    res = m.generate_raw(fx['mels'], False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=fx['noise']['expo'])
    assert m.last_timing['kernel'] == _cabi.KERNEL_TEAM2 and m.last_timing['launches'] >= 10
    got = res['labels'].cpu().numpy().T
    assert got.shape == (401 * 275, 1)
    om = orc.OracleModel(fx['state_dict'], fast=True)
    cm, ca = om.conditioning(fx['mels'])
    ref = om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'])
    first = check_free_run_raw(got, ref)
    st = label_stats(got, ref['labels'], first)
    print(f'\n[parity configs[1] injected] steps compared {st["compared"]}, mismatches {st["mismatches"]}, '
          f'max |dlabel| {st["max_abs"]}, first near-tie divergence {first}')
    assert st['max_abs'] <= 1
    if first[0] is None:
        # identical to the oracle over the whole clip => must equal what the reference itself produced
        np.testing.assert_array_equal(got, fx['labels'].astype(np.int32))
        wav = m.generate(fx['mels'], '/tmp/wrnn_c1.wav', False, 11000, 550, True, noise_mode=_cabi.NOISE_INJECTED,
                         noise1=fx['noise']['expo'])
        assert wav.dtype == np.float64 and wav.shape == (400 * 275,)
        np.testing.assert_array_equal(wav.astype(np.float32), fx['wav'])


def test_config1_b1_t401_philox_production_mode():
    """configs[1] exactly as bench.py runs it (device Philox noise): the draws are reproduced on the host and fed to
    the oracle."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1000, 1, 401)
    m = _model(sd)
    seed = 0xC0FFEE
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=seed)
    got = res['labels'].cpu().numpy().T
    L = got.shape[0]
    q = _philox_q(seed, L, [0])
    om = orc.OracleModel(sd, fast=True)
    cm, ca = om.conditioning(mels)
    ref = om.loop(cm, ca, orc.NOISE_EXPO, q)
    first = check_free_run_raw(got, ref)
    st = label_stats(got, ref['labels'], first)
    print(f'\n[parity configs[1] philox] steps compared {st["compared"]}, mismatches {st["mismatches"]}, '
          f'max |dlabel| {st["max_abs"]}, first near-tie divergence {first}')
    assert st['max_abs'] <= 1
    assert len(np.unique(got)) > 100


def _philox_q(seed, L, rows, chunk=4096):
    """Exp(1) draws q = -log(u) of the device's Philox stream for steps [0, L) and the given GLOBAL rows, replayed with torch
    on the GPU (float64 log), as a float32 numpy array (L, len(rows), 1024) for the oracle."""
    from tests.philox_ref import philox_uniform_raw_torch
    q = np.empty((L, len(rows), 1024), np.float32)
    for t0 in range(0, L, chunk):
        n = min(chunk, L - t0)
        u = philox_uniform_raw_torch(seed, t0, n, rows, device='cuda')
        q[t0:t0 + n] = (-torch.log(u.to(torch.float64))).to(torch.float32).cpu().numpy()
    return q


def test_config2_b64_sampled_batch_kernel():
    """configs[2]: 64 distinct utterances, sampled (not greedy) with the production Philox noise, T=41 (11 275 steps):
    AUTO picks the batch kernel (8 rows per XCD team in lock-step on the matrix cores).  8 of the 64 rows -- one per
    team, different positions inside a team's row octet -- are checked against the oracle with the host replay of
    their draws (the noise is keyed by the global row index, rows are independent)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    B, T = 64, 41
    mels = make_mels(4321, B, T)
    m = _model(sd)
    seed = 0x5EED0064
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=seed)
    assert m.last_timing['kernel'] == _cabi.KERNEL_BATCH
    lab = res['labels'].cpu().numpy()          # (64, L)
    L = lab.shape[1]
    assert lab.shape == (B, T * 275)
    rows = [0, 9, 18, 27, 36, 45, 54, 63]
    q = _philox_q(seed, L, rows)
    om = orc.OracleModel(sd, fast=True)
    cm, ca = om.conditioning(mels[rows])
    ref = om.loop(cm, ca, orc.NOISE_EXPO, q)
    got = lab[rows].T
    first = check_free_run_raw(got, ref)
    st = label_stats(got, ref['labels'], first)
    print(f'\n[parity configs[2] B=64 philox, batch kernel] rows {rows}: steps compared {st["compared"]}, mismatches '
          f'{st["mismatches"]}, max |dlabel| {st["max_abs"]}, first near-tie divergence {first}')
    assert st['max_abs'] <= 1
    # all 64 rows are different utterances with different noise: no two label sequences coincide
    assert len({lab[i, :2000].tobytes() for i in range(B)}) == B
    smp = res['samples'].cpu().numpy()
    np.testing.assert_array_equal(smp, 2.0 * lab.astype(np.float32) / np.float32(1023.0) - np.float32(1.0))


def test_config4_mol_b32_batch_kernel():
    """configs[4]: MOL 9-bit, B=32, injected u_mix / u_log, T=41; the batch kernel (4 rows per team).  Mixture index
    identical (near-tie rule) and the continuous sample within 1/4 of a 9-bit LSB in free-running mode."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, mode='MOL', variant='default', bits=9)
    B, T = 32, 41
    L = T * 275
    mels = make_mels(777, B, T)
    rng = np.random.Generator(np.random.PCG64(2024))
    u_mix = rng.uniform(1e-5, 1.0 - 1e-5, size=(L, B, 10)).astype(np.float32)
    u_log = rng.uniform(1e-5, 1.0 - 1e-5, size=(L, B)).astype(np.float32)
    m = _model(sd, mode='MOL', bits=9)
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=u_mix, noise2=u_log)
    assert m.last_timing['kernel'] == _cabi.KERNEL_BATCH
    smp, mix = res['samples'].cpu().numpy(), res['labels'].cpu().numpy()
    rows = [0, 5, 10, 15, 20, 25, 30, 31]
    om = orc.OracleModel(sd, mode='MOL', bits=9, fast=True)
    ref = _oracle_rows(om, mels, rows, 0, u_mix, u_log)
    check_mol(smp[rows].T, mix[rows].T, ref, teacher_forced=False)
    same = mix[rows].T == ref['labels']
    err = np.abs(smp[rows].T - ref['samples'])[same]
    print(f'\n[parity configs[4] MOL B=32, batch kernel] rows {rows}: mixture index equal on {same.mean() * 100:.3f} % of steps, '
          f'max |sample error| {err.max():.3e} = {err.max() / MOL_LSB:.4f} LSB(9 bit)')
    assert np.abs(smp).max() <= 1.0


@pytest.mark.parametrize('kernel', ['team2', 'batch'])
def test_b8_t60_reference_golden(kernel):
    """A reference-minted golden long enough for several natural segments per row at B = 8 (16 500 steps, no developer
    knob): the latency kernel (one row per XCD team, 8 default segments) and the batch kernel (forced: 8 teams x 1 row
    would be AUTO's choice) against the labels of the unmodified reference."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    fx = load_case('raw_peaky_b8_t60')
    m = _model(fx['state_dict'], kernel=kernel)
    res = m.generate_raw(fx['mels'], False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=fx['noise']['expo'])
    if kernel == 'team2':
        assert m.last_timing['launches'] >= 4
    got = res['labels'].cpu().numpy().T
    om = orc.OracleModel(fx['state_dict'], fast=True)
    cm, ca = om.conditioning(fx['mels'])
    ref = om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'])
    np.testing.assert_array_equal(ref['labels'], fx['labels'].astype(np.int32))   # the oracle reproduces the reference here too
    first = check_free_run_raw(got, ref)
    st = label_stats(got, ref['labels'], first)
    print(f'\n[parity B=8 T=60 {kernel}] steps compared {st["compared"]}, mismatches {st["mismatches"]}, max |dlabel| {st["max_abs"]}, '
          f'first near-tie divergence {first}')
    assert st['max_abs'] <= 1
    if all(f is None for f in first):
        np.testing.assert_array_equal(got, fx['labels'].astype(np.int32))


@pytest.mark.parametrize('kernel', ['team2', 'batch', 'simple'])
def test_raw_9bit_has_no_phantom_classes(kernel):
    """RAW with bits < 10 (n_classes < 1024): workgroups / quarter-waves that own no class must not enter the race.
    Labels stay below n_classes and match the oracle."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    bits = 9
    sd = make_state_dict(3, variant='peaky', bits=bits)
    B = 5 if kernel != 'team2' else 2
    mels = make_mels(55, B, 21)
    L = 21 * 275
    rng = np.random.Generator(np.random.PCG64(9))
    q = rng.standard_exponential((L, B, 512)).astype(np.float32)
    m = _model(sd, bits=bits, kernel=kernel)
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=q)
    got = res['labels'].cpu().numpy().T
    assert got.min() >= 0 and got.max() < 512
    om = orc.OracleModel(sd, bits=bits, fast=True)
    cm, ca = om.conditioning(mels)
    ref = om.loop(cm, ca, orc.NOISE_EXPO, q)
    first = check_free_run_raw(got, ref)
    smp = res['samples'].cpu().numpy().T
    np.testing.assert_array_equal(smp, 2.0 * got.astype(np.float32) / np.float32(511.0) - np.float32(1.0))
    assert label_stats(got, ref['labels'], first)['max_abs'] <= 1


def test_generate_many_fills_the_teams_with_ragged_utterances(tmp_path):
    """Serving extension: 5 utterances of different lengths in one device call (one per XCD team on the latency kernel).
    Row i, trimmed to its own length, must be what a single generate() call on clip i gives for the same noise."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    m = _model(sd)
    lens = [21, 33, 27, 40, 22]
    clips = [make_mels(300 + i, 1, t)[0] for i, t in enumerate(lens)]
    tmax = max(lens)
    rng = np.random.Generator(np.random.PCG64(77))
    q = rng.standard_exponential((tmax * 275, len(lens), 1024)).astype(np.float32)
    outs = m.generate_many(clips, [tmp_path / f'{i}.wav' for i in range(len(lens))], True, noise_mode=_cabi.NOISE_INJECTED, noise1=q)
    assert m.last_timing['kernel'] == _cabi.KERNEL_TEAM2 and m.last_timing['rows'] == len(lens)
    for i, t in enumerate(lens):
        assert outs[i].shape == ((t - 1) * 275,) and outs[i].dtype == np.float64
        single = m.generate(clips[i][None], tmp_path / 's.wav', False, 11000, 550, True, noise_mode=_cabi.NOISE_INJECTED,
                            noise1=np.ascontiguousarray(q[:t * 275, i:i + 1]))
        np.testing.assert_array_equal(outs[i], single)
        assert (tmp_path / f'{i}.wav').exists()
