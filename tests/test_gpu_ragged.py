"""Ragged batches (wrnn_sample_opts.frames_dev, ABI 4): utterances of different lengths in one device call.

The reference has no such call (its batch is one tensor, fatchord_version.py:183); rows are independent there
(:194-196), so the contract is: the first frames[b] * hop outputs of row b are what the padded call -- and a call on that clip
alone -- produce, the rest of the row is left unwritten, and a short clip costs its own length on the device.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _model(kernel):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    sd = make_state_dict(0, variant='peaky')
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.kernel = _cabi.KERNEL_IDS[kernel]
    return m, sd


def _padded(lens, seed):
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    tmax = max(lens)
    batch = np.zeros((len(lens), 80, tmax), np.float32)
    for i, t in enumerate(lens):
        batch[i, :, :t] = make_mels(seed + i, 1, t)[0]
    return batch


@pytest.mark.parametrize('kernel,n', [('batch', 19), ('batch', 70), ('batch_cs', 19), ('batch_cs', 70), ('team2', 11), ('simple', 3)])
def test_ragged_rows_equal_the_padded_call_and_stop_at_their_own_length(kernel, n):
    """Greedy sampling (no noise arrays): a ragged call against the padded call on the same kernel -- the valid part of every
    row bit-equal, nothing written past a row's own length -- and row 0..2 against the oracle."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    m, sd = _model(kernel)
    rng = np.random.Generator(np.random.PCG64(n))
    lens = [int(t) for t in rng.integers(4, 24, size=n)]
    lens[n // 2] = 24                                      # the longest clip sits in the middle of the batch
    mels = _padded(lens, 500)
    pad = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX)
    rag = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX, frames=np.asarray(lens, np.int32))
    lp, lr = pad['labels'].cpu().numpy(), rag['labels'].cpu().numpy()
    sr = rag['samples'].cpu().numpy()
    assert lr.shape == (n, 24 * 275)
    for i, t in enumerate(lens):
        np.testing.assert_array_equal(lr[i, :t * 275], lp[i, :t * 275], err_msg=f'row {i} (T={t})')
        assert not lr[i, t * 275:].any() and not sr[i, t * 275:].any(), f'row {i}: written past its own length'
    om = orc.OracleModel(sd, fast=True)
    for i in range(3):
        cm, ca = om.conditioning(mels[i:i + 1, :, :lens[i]])
        ref = om.loop(cm, ca, orc.NOISE_ARGMAX)
        from tests.parity_util import check_free_run_raw
        check_free_run_raw(lr[i:i + 1, :lens[i] * 275].T, ref)


def test_ragged_batch_costs_the_sum_of_its_lengths_not_rows_times_the_longest():
    """128 clips, 21..80 frames, on the batch kernel (8 teams x 8 rows = 64 rows per pass, two passes).  Padded: every pass runs
    80 frames.  Ragged: the rows are sorted on the device, batches hold rows of similar length and are dealt to the teams in
    snake order -- the loop kernel must be clearly faster (expected ~0.65x; asserted < 0.85x)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    m, _ = _model('batch')
    rng = np.random.Generator(np.random.PCG64(128))
    lens = [int(t) for t in rng.integers(21, 81, size=128)]
    lens[5] = 80
    mels = _padded(lens, 900)
    m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=1)   # warm-up (allocations)
    m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=1)
    t_pad = m.last_timing['loop_ms']
    m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=1, frames=np.asarray(lens, np.int32))
    t_rag = m.last_timing['loop_ms']
    print(f'\n[ragged] 128 clips, sum T = {sum(lens)} vs 128 x 80 = {128 * 80}: loop {t_rag:.1f} ms ragged, {t_pad:.1f} ms padded '
          f'({t_rag / t_pad:.2f}x)')
    assert t_rag < 0.85 * t_pad


def test_generate_many_device_epilogue_equals_host_epilogue(tmp_path):
    """generate_many(epilogue='device') = one wrnn_epilogue_rows launch for all clips, ragged lengths: the same float64
    waveforms as the NumPy tail (same tolerance as wrnn_epilogue: the host libm's pow)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    m, _ = _model('auto')
    lens = [21, 40, 33, 25, 21, 38, 29, 22, 36, 31]
    clips = [make_mels(40 + i, 1, t)[0] for i, t in enumerate(lens)]
    host = m.generate_many(clips, None, True, noise_mode=_cabi.NOISE_PHILOX, seed=5)
    dev = m.generate_many(clips, [tmp_path / f'{i}.wav' for i in range(len(lens))], True, epilogue='device',
                          noise_mode=_cabi.NOISE_PHILOX, seed=5)
    assert m.last_timing['kernel'] in (_cabi.KERNEL_BATCH, _cabi.KERNEL_BATCH_CS)
    for i, t in enumerate(lens):
        assert dev[i].shape == ((t - 1) * 275,) and dev[i].dtype == np.float64
        np.testing.assert_allclose(dev[i], host[i], rtol=0, atol=4 * np.finfo(np.float64).eps)
        assert (tmp_path / f'{i}.wav').exists()
    with pytest.raises(ValueError):   # a clip shorter than the 20-hop fade-out: the reference's broadcast error (:258)
        m.generate_many([clips[0][:, :20]], None, True)


def test_two_handles_share_a_gpu_without_timeouts():
    """Co-residency is enforced, not a rule for the caller: a RAW and a MOL model (two handles) launching team kernels from two
    host threads on two streams at the same time are ordered behind each other on the device (per-device launch gate) -- both
    finish, with the results of a solo run."""
    import threading
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    m1, _ = _model('auto')
    sd2 = make_state_dict(0, mode='MOL', variant='default', bits=9)
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = 9
    m2 = WaveRNN(**dims, mode='MOL')
    m2.verbose = False
    m2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
    m2.to('cuda:0')
    mels1, mels2 = make_mels(1, 3, 30), make_mels(2, 12, 30)
    solo1 = m1.generate_raw(mels1, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=3)['labels'].cpu().numpy()
    solo2 = m2.generate_raw(mels2, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=4)['samples'].cpu().numpy()
    out, errs = {}, []

    def run(key, model, mels, seed, field):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    out[key] = model.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=seed)[field].cpu().numpy()
        except Exception as e:  # surfaced below
            errs.append(e)
    th = [threading.Thread(target=run, args=('a', m1, mels1, 3, 'labels')), threading.Thread(target=run, args=('b', m2, mels2, 4, 'samples'))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    np.testing.assert_array_equal(out['a'], solo1)
    np.testing.assert_array_equal(out['b'], solo2)


@pytest.mark.parametrize('mode,kernel,B', [('RAW', 'team2', 3), ('RAW', 'batch', 12), ('RAW', 'batch', 40), ('MOL', 'batch', 12), ('RAW', 'batch_cs', 12), ('MOL', 'batch_cs', 12)])
def test_phase_profile_runs_the_instrumented_kernels_with_the_same_results(mode, kernel, B):
    """wrnn_phase_profile / wrnn_phase_cycles (ABI 4; ABI 3 read an environment variable): the instrumented instantiations of the
    team kernels ship in the library, so they are tested like everything else -- same labels / samples as the plain kernel,
    plausible cycle counts (the instrumented RAW batch kernel used to fault: its 24 accumulators per wave now live in LDS)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    bits = 10 if mode == 'RAW' else 9
    sd = make_state_dict(0, mode=mode, variant='peaky' if mode == 'RAW' else 'default', bits=bits)
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = bits
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.kernel = _cabi.KERNEL_IDS[kernel]
    mels = make_mels(3, B, 6)
    nat = m.native()
    plain = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=9)
    with pytest.raises(_cabi.WrnnError):
        nat.phase_cycles()                                   # nothing instrumented ran yet
    nat.phase_profile(True)
    prof = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=9)
    cyc = nat.phase_cycles()
    nat.phase_profile(False)
    np.testing.assert_array_equal(prof['labels'].cpu().numpy(), plain['labels'].cpu().numpy())
    np.testing.assert_array_equal(prof['samples'].cpu().numpy(), plain['samples'].cpu().numpy())
    assert cyc.shape == (8, 32)
    if not (mode == 'MOL' and kernel == 'team2'):            # the latency kernel's instrumented build exists for RAW only
        total = cyc[0].sum()
        assert 3000 < total < 200000, total                  # cycles per step of wave 0: a few us at ~2.4 GHz
    again = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=9)
    np.testing.assert_array_equal(again['labels'].cpu().numpy(), plain['labels'].cpu().numpy())
