"""Round 6: single-utterance latency mode, the reference's own noise stream through the product surface, and the loud slow path.

* Folded ("batched") generation of ONE utterance at the BASELINE clip size (mel 80x401) on the kernels `target='auto'` and the reference's
  hp defaults select: 64 folds on the batch kernel (8 rows per XCD team), and `batched=True, 11000, 550` (wavernn_hparams.py:55-57: 10 folds,
  2 rows per team in half-empty quads through the fold row table, fatchord_version.py:293-340).  EVERY step of EVERY fold against the
  oracle's `fold` + loop driven along the GPU's own trajectory, and the crossfaded / unfolded float64 waveform against the oracle's
  `epilogue` (= xfade_and_unfold :342-405 + tail :255-258).
* `noise_mode='reference'`: `torch.manual_seed(42); m.generate(...)` must be the reference-minted golden wav -- the draws come from
  `vocoder.reference_noise` (product code), NOT from oracle/noise.py; this test loads the fixtures without importing `oracle`.
* AUTO on the any-shape kernel warns (once, with the library's reason); WRNN_ERR_BUSY is retried once before falling back.
"""
import os
import warnings

import numpy as np
import pytest
import torch

from tests.parity_util import MOL_LSB, bound_near_ties, check_on_gpu_trajectory_mol, check_on_gpu_trajectory_raw, parity_report

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


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
    m.kernel = _cabi.KERNEL_IDS[kernel]
    return m


def _check_folds_raw(tag, m, sd, mels, target, overlap, seed, want_rows, want_kernel, group=16):
    """One utterance, folded, Philox noise: every step of every fold vs the oracle (fold conditioning, the host replay of the draws keyed by
    the fold's row index), then the device epilogue vs the oracle's float64 tail on the same samples."""
    from oracle import oracle as orc
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tests.test_gpu_baseline_sizes import _philox_q
    res = m.generate_raw(mels, True, target, overlap, noise_mode=_cabi.NOISE_PHILOX, seed=seed)
    assert _cabi.KERNEL_NAMES[m.last_timing['kernel']] == want_kernel
    lab, smp = res['labels'].cpu().numpy(), res['samples'].cpu().numpy()       # (folds, steps)
    rows, steps = lab.shape
    assert rows == want_rows and steps == target + 2 * overlap
    om = orc.OracleModel(sd, fast=True)
    cm, ca = om.conditioning(mels)
    cm, ca = om.fold(cm, target, overlap), om.fold(ca, target, overlap)
    assert cm.shape[0] == rows
    compared, near = 0, []
    for r0 in range(0, rows, group):
        rs = list(range(r0, min(rows, r0 + group)))
        q = _philox_q(seed, steps, rs)
        st = check_on_gpu_trajectory_raw(lab[rs].T, smp[rs].T, lambda xf: om.loop(np.ascontiguousarray(cm[rs]), np.ascontiguousarray(ca[rs]), orc.NOISE_EXPO, q, x_forced=xf))
        compared += st['compared']
        near += [(t, rs[r], d) for t, r, d in st['near_ties']]
    bound_near_ties(tag, compared, near)
    assert compared == rows * steps
    T = mels.shape[-1]
    wave_len = (T - 1) * 275
    want = orc.epilogue(smp, m.n_classes, True, True, target, overlap, wave_len, 275)
    got = m.epilogue_device(res, True, target, overlap, True, wave_len).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=4 * np.finfo(np.float64).eps)
    return rows, steps


def test_fold_auto_64_folds_t401_every_step_and_the_unfolded_wave():
    """`target='auto'` on the configs[1] clip: the cost model's choice (64 folds x 2 265 steps, 8 rows per XCD team on the batch kernel)."""
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import fold_plan
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1000, 1, 401)
    m = _model(sd)
    target = m.fold_target_for_device(401, 550, policy='auto')
    assert target == fold_plan(401 * 275, 550, 8, 'RAW')[0]
    rows, steps = _check_folds_raw('fold auto: 1 utterance x 64 folds, T=401, batch_cs', m, sd, mels, target, 550, 0xF01D, 64, 'batch_cs')
    assert (rows, steps) == (64, 2265)


def test_fold_hp_defaults_t401_every_step_and_the_unfolded_wave():
    """The reference's own fast mode at its own defaults (wavernn_hparams.py:55-57: voc_target 11000, voc_overlap 550) on the configs[1]
    clip: 10 folds x 12 100 steps -- more rows than XCD teams, so AUTO runs the batch kernel with 2 rows per team in half-empty quads."""
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1000, 1, 401)
    m = _model(sd)
    _check_folds_raw('fold hp defaults 11000/550: 1 utterance x 10 folds, T=401, batch_cs', m, sd, mels, 11000, 550, 0xD0F, 10, 'batch_cs', group=10)


def test_fold_per_xcd_t401_every_step():
    """`target='per_xcd'` (round 4's 'auto'): one fold per XCD team on the latency kernel, 8 x 14 266 steps."""
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1000, 1, 401)
    m = _model(sd)
    target = m.fold_target_for_device(401, 550, policy='per_xcd')
    _check_folds_raw('fold per_xcd: 1 utterance x 8 folds, T=401, team2', m, sd, mels, target, 550, 0xC0D, 8, 'team2', group=8)


def test_fold_auto_mol_t401_every_step():
    """MOL in fold mode at the cost model's choice (64 folds: the two-quad MOL instantiation), injected uniforms, every step."""
    from oracle import oracle as orc
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, mode='MOL', variant='default', bits=9)
    mels = make_mels(777, 1, 401)
    m = _model(sd, mode='MOL', bits=9)
    target = m.fold_target_for_device(401, 550, policy='auto')
    rows, steps = m.native().plan(1, 401, True, target, 550)
    rng = np.random.Generator(np.random.PCG64(99))
    u_mix = rng.uniform(1e-5, 1.0 - 1e-5, size=(steps, rows, 10)).astype(np.float32)
    u_log = rng.uniform(1e-5, 1.0 - 1e-5, size=(steps, rows)).astype(np.float32)
    res = m.generate_raw(mels, True, target, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=u_mix, noise2=u_log)
    assert m.last_timing['kernel'] == _cabi.KERNEL_BATCH_CS and rows > 32
    smp, mix = res['samples'].cpu().numpy(), res['labels'].cpu().numpy()
    om = orc.OracleModel(sd, mode='MOL', bits=9, fast=True)
    cm, ca = om.conditioning(mels)
    cm, ca = om.fold(cm, target, 550), om.fold(ca, target, 550)
    st = check_on_gpu_trajectory_mol(smp.T, mix.T, lambda xf: om.loop(cm, ca, 0, u_mix, u_log, x_forced=xf))
    parity_report(f'fold auto MOL: 1 utterance x {rows} folds x {steps} steps, batch_cs: steps compared {st["compared"]}, mixture-index near-ties '
                  f'{st["index_mismatches"]}, max |sample error| {st["max_err"]:.3e} = {st["max_err"] / MOL_LSB:.5f} LSB(9 bit)')
    assert st['compared'] == rows * steps and st['index_mismatches'] <= 1
    wave_len = 400 * 275
    want = orc.epilogue(smp, m.n_classes, False, True, target, 550, wave_len, 275)
    got = m.epilogue_device(res, True, target, 550, False, wave_len).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=4 * np.finfo(np.float64).eps)


def test_target_auto_is_the_cost_models_choice_and_per_xcd_is_one_fold_per_team(tmp_path):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import fold_plan
    sd = make_state_dict(0, variant='peaky')
    m = _model(sd)
    T, overlap = 120, 200
    mels = make_mels(3, 1, T)
    ok, n_teams, why = m.native().team_info()
    assert ok and why == '' and n_teams == torch.cuda.get_device_properties(0).multi_processor_count // 32
    target, folds, _ = fold_plan(T * 275, overlap, n_teams, 'RAW')
    wav = m.generate(mels, tmp_path / 'a.wav', True, 'auto', overlap, True, seed=11, epilogue='device')
    assert (m.last_timing['rows'], m.last_timing['steps']) == (folds, target + 2 * overlap)
    wav2 = m.generate(mels, tmp_path / 'b.wav', True, target, overlap, True, seed=11, epilogue='device')
    np.testing.assert_array_equal(wav, wav2)
    m.generate(mels, tmp_path / 'c.wav', True, 'per_xcd', overlap, True, seed=11)
    assert m.last_timing['rows'] == n_teams and m.last_timing['kernel'] == _cabi.KERNEL_TEAM2
    with pytest.raises(ValueError):
        m.generate(mels, tmp_path / 'd.wav', True, 'fastest', overlap, True)


def _golden(name):
    """A reference-minted fixture and its seeded inputs, WITHOUT oracle/ (tests/golden_util.py replays the noise through oracle.noise)."""
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    fx = {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
    fx['state_dict'] = make_state_dict(int(fx['weight_seed']), mode=fx['mode'], variant=fx['variant'], bits=int(fx['bits']))
    fx['mels'] = make_mels(int(fx['mel_seed']), int(fx['B']), int(fx['T']))
    return fx


_RNG_OK = None


def _same_rng_stream_as_the_minting_build():
    """The goldens pin torch's CPU generator stream of the build they were minted with (a checksum of the draws in every fixture); on a box whose torch
    draws differently the comparison below would be against another noise sequence: skip, like tests/golden_util.load_case does."""
    global _RNG_OK
    if _RNG_OK is None:
        from tacotronv2_wavernn_chinese_amd.vocoder import reference_noise
        z = np.load(os.path.join(GOLDEN_DIR, 'raw_peaky_b1_t24.npz'))
        state = torch.get_rng_state()
        torch.manual_seed(int(z['noise_seed']))
        q = reference_noise('RAW', 1, 24 * 275, 1024, 512, 32, 'cpu')[0].numpy()
        torch.set_rng_state(state)
        got = np.concatenate([q.reshape(-1)[:8].astype(np.float64), [q.astype(np.float64).sum()]])   # = oracle.noise.noise_checksum of {'expo': q}
        _RNG_OK = bool(np.allclose(got, z['noise_checksum'], rtol=0, atol=0))
    return _RNG_OK


@pytest.mark.parametrize('name', ['raw_peaky_b1_t24', 'raw_peaky_fold_t30', 'raw_peaky_b3_t21', 'mol_default_b1_t24', 'mol_default_b2_t21', 'raw_peaky_b1_t401'])
def test_reference_noise_mode_reproduces_the_reference(name, tmp_path):
    """`torch.manual_seed(s); generate(..., noise_mode='reference')` IS the reference's `generate` for seed s: the draws of the global CPU
    generator are replayed by the PRODUCT (vocoder.reference_noise), in the reference's order (:178-179, :231-235 / distribution.py:106,118).
    Compared with the wav (and labels) the unmodified reference produced (oracle/make_golden.py, NOISE_SEED = 42)."""
    if not _same_rng_stream_as_the_minting_build():
        pytest.skip('torch CPU RNG stream differs from the one the goldens were minted with')
    fx = _golden(name)
    m = _model(fx['state_dict'], mode=fx['mode'], bits=int(fx['bits']))
    args = (fx['mels'], tmp_path / 'o.wav', bool(fx['batched']), int(fx['target']), int(fx['overlap']), True)
    torch.manual_seed(int(fx['noise_seed']))
    wav = m.generate(*args, noise_mode='reference')
    assert wav.dtype == np.float64
    if fx['mode'] == 'RAW':
        torch.manual_seed(int(fx['noise_seed']))
        res = m.generate_raw(fx['mels'], bool(fx['batched']), int(fx['target']), int(fx['overlap']), noise_mode='reference')
        lab = res['labels'].cpu().numpy().T
        nbad = int(np.count_nonzero(lab != fx['labels'].astype(np.int32)))
        parity_report(f"noise_mode='reference' {name}: {lab.size} steps, labels differing from the reference's own {nbad}")
        # bit-equal labels and wav (these fixtures have no near-tie on the shipped kernels: test_config1_*, test_raw_free_running_*)
        assert nbad == 0
        ref_wav = fx['wav']
        np.testing.assert_array_equal(wav.astype(ref_wav.dtype), ref_wav)     # the T=401 fixture stores the wav as float32
    else:
        np.testing.assert_allclose(wav, fx['wav'], rtol=0, atol=1e-4)           # continuous fed-back value: fp32 round-off accumulates
    # and the mode is reproducible / seed-sensitive like the reference
    torch.manual_seed(int(fx['noise_seed']) + 1)
    other = m.generate(*args, noise_mode='reference')
    assert not np.array_equal(other, wav)


def test_auto_on_the_any_shape_kernel_warns_once_with_the_reason(tmp_path):
    """Default dims, team kernels unavailable (test hook = what a failed residency check gives): AUTO runs WRNN_KERNEL_SIMPLE and says so."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    m = _model(sd)
    mels = make_mels(9, 1, 21)
    fast = m.generate(mels, tmp_path / 'f.wav', False, 11000, 550, True, seed=3)
    nat = m.native()
    nat.debug_force_no_teams(True)
    assert nat.team_info()[0] is False and 'test hook' in nat.team_info()[2]
    with pytest.warns(RuntimeWarning, match='WRNN_KERNEL_SIMPLE.*test hook'):
        slow = m.generate(mels, tmp_path / 's.wav', False, 11000, 550, True, seed=3)
    assert m.last_timing['kernel'] == _cabi.KERNEL_SIMPLE
    np.testing.assert_array_equal(slow, fast)          # same Philox draws, same labels (T = 21: no near-tie between the two kernels here)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                  # once per model
        m.generate(mels, tmp_path / 's.wav', False, 11000, 550, True, seed=4)
    with pytest.raises(_cabi.WrnnError):               # an explicit team kernel is an error, not a silent fallback
        m.generate(mels, tmp_path / 's.wav', False, 11000, 550, True, seed=4, kernel=_cabi.KERNEL_TEAM2)
    nat.debug_force_no_teams(False)
    m.generate(mels, tmp_path / 's.wav', False, 11000, 550, True, seed=4)
    assert m.last_timing['kernel'] == _cabi.KERNEL_TEAM2


def test_busy_gpu_is_retried_once_then_falls_back_loudly(tmp_path, monkeypatch):
    """WRNN_ERR_BUSY (the team kernel's workgroups did not all become resident: another process holds CUs) under AUTO: one retry after
    `busy_retry_seconds`; busy again -> the any-shape kernel with the warning.  The error is injected at the binding (a second process
    holding CUs is not something a test can arrange reliably)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    m = _model(sd)
    m.busy_retry_seconds = 0.01
    mels = make_mels(9, 1, 21)
    want = m.generate(mels, tmp_path / 'a.wav', False, 11000, 550, True, seed=8)
    nat = m.native()
    real = nat.last_timing
    fails = {'n': 1}

    def flaky():
        t = real()
        if fails['n'] > 0 and t['kernel'] != _cabi.KERNEL_SIMPLE:
            fails['n'] -= 1
            raise _cabi.WrnnError(_cabi.ERR_BUSY, 'injected')
        return t
    monkeypatch.setattr(nat, 'last_timing', flaky)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                  # one BUSY: retried, no fallback, no warning
        got = m.generate(mels, tmp_path / 'b.wav', False, 11000, 550, True, seed=8)
    np.testing.assert_array_equal(got, want)
    assert m.last_timing['kernel'] == _cabi.KERNEL_TEAM2
    fails['n'] = 2
    with pytest.warns(RuntimeWarning, match='shared with another kernel'):
        got = m.generate(mels, tmp_path / 'c.wav', False, 11000, 550, True, seed=8)
    assert m.last_timing['kernel'] == _cabi.KERNEL_SIMPLE
    np.testing.assert_array_equal(got, want)
    fails['n'] = 1
    with pytest.raises(_cabi.WrnnError):               # an explicit kernel request is not second-guessed
        m.generate(mels, tmp_path / 'd.wav', False, 11000, 550, True, seed=8, kernel=_cabi.KERNEL_TEAM2)
