"""Parity of the HIP path (through the C-ABI) against the oracle and the reference's golden vectors.
Runs on the MI355X box only (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.golden_util import ALL_CASES, MOL_CASES, RAW_CASES, load_case
from tests.parity_util import check_free_run_raw, check_mol, check_teacher_forced_raw

pytestmark = pytest.mark.gpu

KERNELS = ['team2', 'batch', 'batch_cs', 'simple']
# The straightforward one-workgroup kernel runs ~1 ms/step: it is exercised on the B=3 case (3 rows in
# parallel) and on the fold case only; the team kernel (the shipped path) runs every case.
SIMPLE_CASES = {'raw_peaky_b3_t21', 'raw_peaky_fold_t30', 'mol_default_b2_t21'}


def _skip_slow(name, kernel):
    if kernel == 'simple' and name not in SIMPLE_CASES:
        pytest.skip('simple kernel: covered on the multi-row cases only (1 ms/step)')


def _model(fx, kernel='auto'):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = int(fx['bits'])
    m = WaveRNN(**dims, mode=fx['mode'])
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in fx['state_dict'].items()})
    m.to('cuda:0')
    m.kernel = _cabi.KERNEL_IDS[kernel]
    return m


_ORACLE_CACHE = {}


def _oracle(fx, x_forced=None, want_logits=False):
    """Oracle runs are cached per (case, free/forced): they are CPU seconds the GPU box pays for."""
    key = (fx['name'], x_forced is not None)
    hit = _ORACLE_CACHE.get(key)
    if hit is not None and (hit['logits'] is not None or not want_logits):
        return hit
    out = _oracle_run(fx, x_forced, True)
    _ORACLE_CACHE[key] = out
    return out


def _oracle_run(fx, x_forced=None, want_logits=False):
    # OpenMP/AVX2 build of the same C restatement (sums re-associated): the comparison rules already allow
    # for summation-order differences, and the GPU box pays for every CPU second
    om = orc.OracleModel(fx['state_dict'], mode=fx['mode'], bits=int(fx['bits']), fast=True)
    cm, ca = om.conditioning(fx['mels'])
    if fx['batched']:
        cm = om.fold(cm, int(fx['target']), int(fx['overlap']))
        ca = om.fold(ca, int(fx['target']), int(fx['overlap']))
    if fx['mode'] == 'RAW':
        return om.loop(cm, ca, orc.NOISE_EXPO, fx['noise']['expo'], x_forced=x_forced, want_logits=want_logits)
    return om.loop(cm, ca, 0, fx['noise']['u_mix'], fx['noise']['u_log'], x_forced=x_forced, want_logits=want_logits)


def _noise_kwargs(fx):
    from tacotronv2_wavernn_chinese_amd import _cabi
    if fx['mode'] == 'RAW':
        return dict(noise_mode=_cabi.NOISE_INJECTED, noise1=fx['noise']['expo'])
    return dict(noise_mode=_cabi.NOISE_INJECTED, noise1=fx['noise']['u_mix'], noise2=fx['noise']['u_log'])


@pytest.mark.parametrize('name', ALL_CASES)
def test_conditioning_matches_reference(name):
    """Rows A2-A5: pad + MelResNet + (5,5,11) upsample vs the reference's own tensors."""
    fx = load_case(name)
    m = _model(fx)
    nat = m.native()
    B, F, T = fx['mels'].shape
    L = T * 275
    mels = torch.from_numpy(fx['mels']).cuda()
    up = torch.empty((B, L, 80), device='cuda')
    aux = torch.empty((B, L, 128), device='cuda')
    nat.conditioning(mels.data_ptr(), B, T, up.data_ptr(), aux.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    up, aux = up.cpu().numpy(), aux.cpu().numpy()
    np.testing.assert_allclose(up[:, :320], fx['up_head'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(up[:, -320:], fx['up_tail'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(up[:, ::41], fx['up_stride'], rtol=0, atol=3e-6)
    np.testing.assert_allclose(aux[:, ::275], fx['aux_frames'], rtol=0, atol=2e-5)
    # nearest-neighbour stretch of aux: constant inside a frame
    np.testing.assert_array_equal(aux[:, 0:275], np.repeat(aux[:, :1], 275, axis=1))


@pytest.mark.parametrize('kernel', KERNELS)
@pytest.mark.parametrize('name', RAW_CASES)
def test_raw_teacher_forced_every_step(name, kernel):
    _skip_slow(name, kernel)
    fx = load_case(name)
    free = _oracle(fx)
    # the reference's own labels are the golden truth for the fed-back sequence
    np.testing.assert_array_equal(free['labels'], fx['labels'].astype(np.int32))
    ref = _oracle(fx, x_forced=free['samples'], want_logits=True)
    m = _model(fx, kernel)
    res = m.generate_raw(fx['mels'], bool(fx['batched']), int(fx['target']), int(fx['overlap']),
                         x_forced=free['samples'], want_logits=True, **_noise_kwargs(fx))
    got = res['labels'].cpu().numpy().T
    nbad = check_teacher_forced_raw(got, ref)
    assert nbad <= max(2, got.size // 2000)
    lg = res['logits'].cpu().numpy()
    scale = max(1.0, float(np.abs(ref['logits']).max()))
    assert np.abs(lg - ref['logits']).max() <= 2e-5 * scale


@pytest.mark.parametrize('kernel', KERNELS)
@pytest.mark.parametrize('name', RAW_CASES)
def test_raw_free_running_matches_reference_labels(name, kernel):
    _skip_slow(name, kernel)
    fx = load_case(name)
    ref = _oracle(fx)
    m = _model(fx, kernel)
    res = m.generate_raw(fx['mels'], bool(fx['batched']), int(fx['target']), int(fx['overlap']), **_noise_kwargs(fx))
    got = res['labels'].cpu().numpy().T
    first = check_free_run_raw(got, ref)
    # samples are the label mapped to [-1, 1] exactly like fatchord_version.py:235
    smp = res['samples'].cpu().numpy().T
    np.testing.assert_array_equal(smp, (2.0 * got.astype(np.float32) / np.float32(1023.0) - np.float32(1.0)))
    if all(f is None for f in first):
        np.testing.assert_array_equal(got, fx['labels'].astype(np.int32))


@pytest.mark.parametrize('kernel', KERNELS)
@pytest.mark.parametrize('name', MOL_CASES)
def test_mol_parity(name, kernel):
    _skip_slow(name, kernel)
    fx = load_case(name)
    m = _model(fx, kernel)
    ref = _oracle(fx)
    np.testing.assert_allclose(ref['samples'], fx['samples'], rtol=0, atol=2e-6)
    res = m.generate_raw(fx['mels'], False, 11000, 550, **_noise_kwargs(fx))
    check_mol(res['samples'].cpu().numpy().T, res['labels'].cpu().numpy().T, ref, teacher_forced=False)
    reft = _oracle(fx, x_forced=ref['samples'])
    res = m.generate_raw(fx['mels'], False, 11000, 550, x_forced=ref['samples'], **_noise_kwargs(fx))
    check_mol(res['samples'].cpu().numpy().T, res['labels'].cpu().numpy().T, reft, teacher_forced=True)


@pytest.mark.parametrize('name', ['raw_peaky_b3_t21', 'raw_peaky_fold_t30', 'mol_default_b2_t21'])
def test_segmented_launches_carry_the_recurrent_state(name):
    """The shipped kernel runs a row as a sequence of launches (one conditioning-stream chunk each) and hands
    h1/h2/gh1/gh2/x over through device memory: with a tiny segment (dozens of launches per row, several rows
    per team) the result must be what a single launch gives, and what the oracle gives."""
    fx = load_case(name)
    m = _model(fx, 'team2')
    args = (fx['mels'], bool(fx.get('batched', False)), int(fx.get('target', 11000)), int(fx.get('overlap', 550)))
    whole = m.generate_raw(*args, **_noise_kwargs(fx))
    parts = m.generate_raw(*args, team2_segment=96, **_noise_kwargs(fx))   # wrnn_sample_opts.team2_segment (ABI 4)
    assert whole['labels'].shape[1] > 10 * 96
    np.testing.assert_array_equal(parts['labels'].cpu().numpy(), whole['labels'].cpu().numpy())
    np.testing.assert_array_equal(parts['samples'].cpu().numpy(), whole['samples'].cpu().numpy())
    ref = _oracle(fx)
    if fx['mode'] == 'RAW':
        check_free_run_raw(parts['labels'].cpu().numpy().T, ref)
    else:
        check_mol(parts['samples'].cpu().numpy().T, parts['labels'].cpu().numpy().T, ref, teacher_forced=False)


@pytest.mark.parametrize('name', ['raw_peaky_b1_t24', 'raw_peaky_fold_t30', 'raw_peaky_b3_t21', 'mol_default_b1_t24'])
@pytest.mark.parametrize('mu_law', [True, False])
def test_device_epilogue_matches_numpy_float64(name, mu_law):
    """wrnn_epilogue (decode_mu_law + xfade_and_unfold + trim + fade-out on the GPU, float64) against the
    oracle's NumPy restatement of fatchord_version.py:243-258 on the same samples.  Everything but pow() is
    the same IEEE operation sequence; tolerance: 4 ulp of the largest magnitude."""
    fx = load_case(name)
    m = _model(fx, 'team2')
    batched, target, overlap = bool(fx.get('batched', False)), int(fx.get('target', 11000)), int(fx.get('overlap', 550))
    res = m.generate_raw(fx['mels'], batched, target, overlap, **_noise_kwargs(fx))
    T = fx['mels'].shape[-1]
    wave_len = (T - 1) * 275
    mu = mu_law and fx['mode'] == 'RAW'
    want = orc.epilogue(res['samples'].cpu().numpy(), m.n_classes, mu, batched, target, overlap, wave_len, 275)
    got = m.epilogue_device(res, batched, target, overlap, mu, wave_len).cpu().numpy()
    assert got.dtype == np.float64 and got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=4 * np.finfo(np.float64).eps)
    assert np.count_nonzero(got != want) <= got.size // 100     # pow() rounding only, if at all


def test_device_epilogue_rejects_what_the_reference_rejects():
    from tacotronv2_wavernn_chinese_amd._cabi import WrnnError
    fx = load_case('raw_peaky_b1_t24')
    m = _model(fx, 'team2')
    res = m.generate_raw(fx['mels'], False, 11000, 550, **_noise_kwargs(fx))
    with pytest.raises(WrnnError):      # T < 21: shorter than the 20-hop fade-out (:258 raises ValueError)
        m.epilogue_device(res, False, 11000, 550, True, 19 * 275)
    with pytest.raises(WrnnError):      # longer than what was generated
        m.epilogue_device(res, False, 11000, 550, True, res['steps'] + 1)
    with pytest.raises(WrnnError):      # folds of the wrong length
        m.epilogue_device(res, True, 11000, 550, True, 23 * 275)


@pytest.mark.parametrize('name', ['raw_peaky_b1_t24', 'raw_peaky_fold_t30', 'raw_peaky_b3_t21', 'mol_default_b1_t24'])
def test_generate_end_to_end_wav(name, tmp_path):
    """The full drop-in call: generate() return value vs the reference's own wav (float64)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    fx = load_case(name)
    m = _model(fx)
    out = tmp_path / 'o.wav'
    wav = m.generate(fx['mels'], out, bool(fx['batched']), int(fx['target']), int(fx['overlap']), True,
                     **_noise_kwargs(fx))
    assert wav.dtype == np.float64 and wav.shape == fx['wav'].shape
    assert m.training  # the reference leaves the module in train mode (:262)
    assert out.exists()
    ref = _oracle(fx)
    res = m.generate_raw(fx['mels'], bool(fx['batched']), int(fx['target']), int(fx['overlap']), **_noise_kwargs(fx))
    if fx['mode'] == 'RAW':
        same = all(f is None for f in check_free_run_raw(res['labels'].cpu().numpy().T, ref))
        if same:
            np.testing.assert_array_equal(wav, fx['wav'])
            # the same call with the float64 tail on the GPU, against the reference's own wav
            wav_dev = m.generate(fx['mels'], out, bool(fx['batched']), int(fx['target']), int(fx['overlap']), True,
                                 epilogue='device', **_noise_kwargs(fx))
            np.testing.assert_allclose(wav_dev, fx['wav'], rtol=0, atol=4 * np.finfo(np.float64).eps)
    else:
        if (res['labels'].cpu().numpy().T == ref['labels']).all():
            np.testing.assert_allclose(wav, fx['wav'], rtol=0, atol=1e-4)


def test_per_xcd_target_makes_one_fold_per_xcd(tmp_path):
    """target='per_xcd' (extension; round 4-5's 'auto'): the folds of one utterance map one-to-one onto the 32-CU teams; the result is
    what the reference's batched mode gives for that explicit target (oracle epilogue on the same samples).  ('auto' = the cost model's
    choice: tests/test_gpu_fold_latency.py.)"""
    fx = load_case('raw_peaky_fold_t30')
    m = _model(fx)
    T, overlap = 30, 100
    mels = fx['mels'][:, :, :T]
    target = m.fold_target_for_device(T, overlap, policy='per_xcd')
    n_teams = torch.cuda.get_device_properties(0).multi_processor_count // 32
    rows, steps = m.native().plan(1, T, True, target, overlap)
    assert rows == n_teams and steps == target + 2 * overlap
    wav = m.generate(mels, tmp_path / 'a.wav', True, 'per_xcd', overlap, True, seed=11)
    wav2 = m.generate(mels, tmp_path / 'b.wav', True, target, overlap, True, seed=11)
    np.testing.assert_array_equal(wav, wav2)
    assert wav.shape == ((T - 1) * 275,)


def test_quirks():
    fx = load_case('raw_peaky_b1_t24')
    m = _model(fx)
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    # T < 21: the reference dies in the fade-out broadcast (fatchord_version.py:256-258)
    with pytest.raises(ValueError):
        m.generate(make_mels(1, 1, 20), '/tmp/wrnn_q.wav', False, 11000, 550, True)
    # batched needs a single utterance (fold_with_overlap, :338)
    with pytest.raises(_cabi.WrnnError):
        m.generate(make_mels(1, 2, 30), '/tmp/wrnn_q.wav', True, 2000, 200, True)
    # philox mode: reproducible under torch.manual_seed, different across seeds
    torch.manual_seed(5)
    a = m.generate(make_mels(2, 1, 21), '/tmp/wrnn_q.wav', False, 11000, 550, True)
    torch.manual_seed(5)
    b = m.generate(make_mels(2, 1, 21), '/tmp/wrnn_q.wav', False, 11000, 550, True)
    torch.manual_seed(6)
    c = m.generate(make_mels(2, 1, 21), '/tmp/wrnn_q.wav', False, 11000, 550, True)
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, c)


def test_philox_sampling_is_distributionally_correct():
    """Own-RNG production mode: the kernel's draws are reproduced on the host (same Philox, numpy) and
    injected into the oracle -> same labels (near-tie rule)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tests.philox_ref import philox_uniform_raw
    fx = load_case('raw_peaky_b1_t24')
    m = _model(fx)
    seed = 0x1234ABCD5678
    res = m.generate_raw(fx['mels'], False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=seed)
    got = res['labels'].cpu().numpy().T
    L = got.shape[0]
    u = philox_uniform_raw(seed, L, 1, 1024)
    q = (-np.log(u.astype(np.float64))).astype(np.float32)
    om = orc.OracleModel(fx['state_dict'])
    cm, ca = om.conditioning(fx['mels'])
    ref = om.loop(cm, ca, orc.NOISE_EXPO, q)
    check_free_run_raw(got, ref)
    # and the histogram is not degenerate
    assert len(np.unique(got)) > 50


@pytest.mark.parametrize('kernel', ['team2', 'batch', 'batch_cs'])
def test_many_rows_are_scheduled_independently(kernel):
    """BASELINE configs[2]-style batch: 19 rows (two full waves of 8 XCD teams + a ragged tail).  Greedy sampling
    (q == 1), rows 0..18 use 3 distinct mels in rotation: rows with the same mel must produce identical label
    sequences whatever team / XCD / position in the row queue ran them, and row 0 must match the oracle."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    fx = load_case('raw_peaky_b1_t24')
    m = _model(fx, kernel)
    base = make_mels(77, 3, 21)
    mels = np.stack([base[i % 3] for i in range(19)])
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX)
    lab = res['labels'].cpu().numpy()
    assert lab.shape == (19, 21 * 275)
    for i in range(3, 19):
        np.testing.assert_array_equal(lab[i], lab[i % 3])
    assert not np.array_equal(lab[0], lab[1])
    if kernel == 'team2':
        # several rows per team AND several segment launches per row: every (row, segment) hands its own state over
        res2 = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX, team2_segment=160)
        assert m.last_timing['launches'] > 30
        np.testing.assert_array_equal(res2['labels'].cpu().numpy(), lab)
    else:
        # 2 (resp. 8) rows per batch: 10 batches over 8 teams (teams 0 and 1 run two batches one after the other: state
        # re-initialised, tags keep counting) resp. 3 batches of 8 + 8 + 3 rows on the 8-row kernel (wrnn_sample_opts.batch_rows)
        for rpb in (2, 8):
            res2 = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX, batch_rows=rpb)
            np.testing.assert_array_equal(res2['labels'].cpu().numpy(), lab)
    om = orc.OracleModel(fx['state_dict'], fast=True)
    cm, ca = om.conditioning(base[:1])
    ref = om.loop(cm, ca, orc.NOISE_ARGMAX)
    check_free_run_raw(lab[:1].T, ref)


def test_cli_and_gen_from_file_end_to_end(tmp_path):
    """wavernn_gen.py mirror: (T, n_mels) .npy in [0,1] + a reference-format checkpoint -> wav on disk."""
    import subprocess
    import sys
    from scipy.io import wavfile
    fx = load_case('raw_peaky_b1_t24')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ckpt = tmp_path / 'latest_weights.pyt'
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in fx['state_dict'].items()}, ckpt)
    mel = tmp_path / 'mel-000.npy'
    np.save(mel, fx['mels'][0].T)                       # (T, 80), the tacotron_synthesize.py:114-116 format
    r = subprocess.run([sys.executable, os.path.join(root, 'wavernn_gen.py'), '--file', str(mel), '-w', str(ckpt), '-u'],
                       cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = tmp_path / 'wavernn_inference_output' / 'mel-000_gen_NOT_BATCHED_step=0k.wav'
    assert out.exists(), r.stdout[-1000:]
    sr, data = wavfile.read(out)
    assert sr == 22050 and data.dtype == np.float32 and data.shape == ((24 - 1) * 275,)
    assert np.isfinite(data).all() and np.abs(data).max() <= 1.0
    # --noise reference --seed S: the wav the reference's own script produces after torch.manual_seed(S) (the fixture was minted with seed 42)
    out.unlink()
    r = subprocess.run([sys.executable, os.path.join(root, 'wavernn_gen.py'), '--file', str(mel), '-w', str(ckpt), '-u', '--noise', 'reference',
                        '--seed', str(int(fx['noise_seed']))], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sr, data = wavfile.read(out)
    np.testing.assert_array_equal(data, fx['wav'].astype(np.float32))
    # batched flag is honoured (the reference overrides it): folded generation writes the batched file name
    r = subprocess.run([sys.executable, os.path.join(root, 'wavernn_gen.py'), '--file', str(mel), '-w', str(ckpt), '-b',
                        '-t', '2000', '-o', '200'], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / 'wavernn_inference_output' / 'mel-000_gen_batched_target2000_overlap200_step=0k.wav').exists()
    # no -w: the training run's latest checkpoint, logs_wavernn/checkpoints/latest_weights.pyt (wavernn_gen.py:112-117,
    # wavernn/utils/paths.py:11-12) -- the same weights, hence the same first file again (greedy-free: only existence + the path printed)
    ck2 = tmp_path / 'logs_wavernn' / 'checkpoints'
    ck2.mkdir(parents=True)
    os.replace(ckpt, ck2 / 'latest_weights.pyt')
    out.unlink()
    r = subprocess.run([sys.executable, os.path.join(root, 'wavernn_gen.py'), '--file', str(mel), '-u'], cwd=tmp_path, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert str(ck2 / 'latest_weights.pyt') in r.stdout and 'randomly initialised' not in r.stdout and out.exists()


def test_edge_shapes_and_bad_arguments():
    """Smallest clips, frame-boundary lengths and invalid arguments through the C-ABI (no crash, loud errors)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    fx = load_case('raw_peaky_b1_t24')
    om = orc.OracleModel(fx['state_dict'], fast=True)
    for kernel in ('team2', 'batch', 'batch_cs'):
        m = _model(fx, kernel)
        for T in (1, 2, 5):                      # shorter than the 5-frame upsampling support / the fade-out
            mels = make_mels(100 + T, 1, T)
            res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX)
            lab = res['labels'].cpu().numpy()
            assert lab.shape == (1, T * 275)
            cm, ca = om.conditioning(mels)
            check_free_run_raw(lab.T, om.loop(cm, ca, orc.NOISE_ARGMAX))
        # folded mode whose last fold is mostly zero padding ('after' padding of fold_with_overlap)
        mels = make_mels(9, 1, 9)
        res = m.generate_raw(mels, True, 1000, 100, noise_mode=_cabi.NOISE_ARGMAX)
        cm, ca = om.conditioning(mels)
        fm, fa = om.fold(cm, 1000, 100), om.fold(ca, 1000, 100)
        assert res['labels'].shape == (fm.shape[0], 1200)
        check_free_run_raw(res['labels'].cpu().numpy().T, om.loop(fm, fa, orc.NOISE_ARGMAX))
    nat = m.native()
    with pytest.raises(_cabi.WrnnError):
        nat.plan(0, 10, False, 11000, 550)
    with pytest.raises(_cabi.WrnnError):
        nat.plan(1, 0, False, 11000, 550)
    with pytest.raises(_cabi.WrnnError):        # a sequence shorter than one fold
        nat.plan(1, 1, True, 11000, 550)
    with pytest.raises(ValueError):
        m.generate_raw(np.zeros((80, 30), np.float32), False, 11000, 550)
    with pytest.raises(_cabi.WrnnError):        # injected mode without noise pointers
        m.generate_raw(make_mels(1, 1, 3), False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED)
