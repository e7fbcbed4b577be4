"""Constructor generality (fatchord_version.py:93-129 takes any dims; wavernn_hparams.py:36-41 are only defaults): a model
that is NOT the reference hparams runs on the any-shape kernel (AUTO falls back to it), checked against the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.parity_util import check_free_run_raw, check_mol

pytestmark = pytest.mark.gpu

SMALL = dict(rnn_dims=256, fc_dims=384, bits=8, pad=2, upsample_factors=(4, 4, 8), feat_dims=40, compute_dims=64,
             res_out_dims=96, res_blocks=3, hop_length=128, sample_rate=16000)


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_non_default_dims_run_on_the_simple_kernel(mode):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(SMALL)
    sd = make_state_dict(5, mode=mode, variant='default', **dims)
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    B, T = 3, 9
    mels = make_mels(21, B, T, feat_dims=40)
    L = T * 128
    rng = np.random.Generator(np.random.PCG64(8))
    om = orc.OracleModel(sd, mode=mode, bits=8, upsample_factors=(4, 4, 8), pad=2, fast=True)
    cm, ca = om.conditioning(mels)
    assert cm.shape == (B, L, 40)
    if mode == 'RAW':
        q = rng.standard_exponential((L, B, 256)).astype(np.float32)
        res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=q)
        assert m.last_timing['kernel'] == _cabi.KERNEL_SIMPLE        # AUTO: the team kernels are built for the reference hparams
        ref = om.loop(cm, ca, orc.NOISE_EXPO, q)
        check_free_run_raw(res['labels'].cpu().numpy().T, ref)
        assert res['labels'].max().item() < 256
    else:
        u1 = rng.uniform(1e-5, 1 - 1e-5, size=(L, B, 10)).astype(np.float32)
        u2 = rng.uniform(1e-5, 1 - 1e-5, size=(L, B)).astype(np.float32)
        res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_INJECTED, noise1=u1, noise2=u2)
        assert m.last_timing['kernel'] == _cabi.KERNEL_SIMPLE
        ref = om.loop(cm, ca, 0, u1, u2)
        check_mol(res['samples'].cpu().numpy().T, res['labels'].cpu().numpy().T, ref, teacher_forced=False)
    with pytest.raises(_cabi.WrnnError):                              # an explicit team kernel on these dims is an error
        m.generate_raw(mels, False, 11000, 550, kernel=_cabi.KERNEL_TEAM2)
    # the prologue on these dims, against the oracle's tensors
    nat = m.native()
    up = torch.empty((B, L, 40), device='cuda')
    aux = torch.empty((B, L, 96), device='cuda')
    nat.conditioning(torch.from_numpy(mels).cuda().data_ptr(), B, T, up.data_ptr(), aux.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(up.cpu().numpy(), cm, rtol=0, atol=3e-6)
    np.testing.assert_allclose(aux.cpu().numpy(), ca, rtol=0, atol=2e-5)


def test_prologue_with_a_window_above_64_kb_of_lds():
    """compute_dims = 1024 (the accepted maximum): the MelResNet kernel's window is (8 + 4) * 80 + 16 * 1024 floats = 69 KB of dynamic
    LDS, above the 64 KB a launch gets without `hipFuncAttributeMaxDynamicSharedMemorySize` (round-2 advisor finding: accepted by
    wrnn_create, failed at the first launch).  Conditioning against the oracle + a short greedy generation."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(rnn_dims=128, fc_dims=128, bits=8, pad=2, upsample_factors=(4, 4, 8), feat_dims=80, compute_dims=1024,
                res_out_dims=128, res_blocks=2, hop_length=128, sample_rate=16000)
    sd = make_state_dict(6, mode='RAW', variant='default', **dims)
    m = WaveRNN(**dims, mode='RAW')
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    B, T = 2, 6
    mels = make_mels(22, B, T)
    L = T * 128
    om = orc.OracleModel(sd, mode='RAW', bits=8, upsample_factors=(4, 4, 8), pad=2, fast=True)
    cm, ca = om.conditioning(mels)
    nat = m.native()
    up = torch.empty((B, L, 80), device='cuda')
    aux = torch.empty((B, L, 128), device='cuda')
    nat.conditioning(torch.from_numpy(mels).cuda().data_ptr(), B, T, up.data_ptr(), aux.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(up.cpu().numpy(), cm, rtol=0, atol=3e-6)
    np.testing.assert_allclose(aux.cpu().numpy(), ca, rtol=0, atol=5e-4 * max(1.0, float(np.abs(ca).max())))
    res = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_ARGMAX)
    assert m.last_timing['kernel'] == _cabi.KERNEL_SIMPLE
    check_free_run_raw(res['labels'].cpu().numpy().T, om.loop(cm, ca, orc.NOISE_ARGMAX))
