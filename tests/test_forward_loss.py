"""N4 of SURVEY.md 8f: the teacher-forced ``forward`` (fatchord_version.py:131-167) and the losses of the training script
(wavernn_train.py:82,112-121; wavernn/utils/distribution.py:16-84) against goldens minted from the unmodified reference
(``python -m oracle.make_golden forward``: eval-mode forward on seeded weights / padded mels / inputs)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['fwd_raw_peaky_b2_t6', 'fwd_mol_default_b2_t6', 'fwd_raw_peaky_b4_t20', 'fwd_mol_default_b4_t20']


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    fx = {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
    fx['state_dict'] = make_state_dict(int(fx['weight_seed']), mode=fx['mode'], variant=fx['variant'], bits=int(fx['bits']))
    fx['mels'] = make_mels(int(fx['mel_seed']), int(fx['B']), int(fx['T']) + 4)
    return fx


@pytest.mark.parametrize('name', CASES)
def test_oracle_losses_match_the_reference(name):
    """CPU: the numpy restatements of the two losses on the reference's own logits reproduce the reference's loss."""
    fx = _load(name)
    ysub = fx['y'][:, ::int(fx['sub_stride'])]
    if fx['mode'] == 'RAW':
        got = orc.cross_entropy(fx['logits_sub'], ysub)
    else:
        got = orc.discretized_mix_logistic_loss(fx['logits_sub'], ysub)
    assert abs(got - fx['loss_sub']) <= 2e-6 * max(1.0, abs(fx['loss_sub'])), (got, fx['loss_sub'])


def _model(fx, kernel):
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = int(fx['bits'])
    m = WaveRNN(**dims, mode=fx['mode'])
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in fx['state_dict'].items()})
    m.to('cuda:0')
    m.kernel = _cabi.KERNEL_IDS[kernel]
    m.eval()   # the goldens are the reference's eval-mode forward; in train() mode forward() is the differentiable pass (test_train_step.py)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['team2', 'batch', 'batch_cs', 'simple'])
@pytest.mark.parametrize('name', CASES)
def test_forward_matches_the_reference(name, kernel):
    """forward(x, mels) on the loop kernels (fed-back value forced to x, first step fed x[:, 0], pre-padded mels) vs the
    reference's eval-mode forward: fc3 outputs within 2e-5 of the largest magnitude; `step` incremented (:139)."""
    fx = _load(name)
    m = _model(fx, kernel)
    step0 = m.get_step()
    y_hat = m.forward(fx['x'], fx['mels'])
    assert m.get_step() == step0 + 1
    B, L = fx['x'].shape
    assert tuple(y_hat.shape) == (B, L, m.n_classes) and y_hat.dtype == torch.float32 and y_hat.is_cuda
    got = y_hat.cpu().numpy()[:, ::int(fx['sub_stride'])]
    scale = max(1.0, float(np.abs(fx['logits_sub']).max()))
    assert np.abs(got - fx['logits_sub']).max() <= 2e-5 * scale
    with pytest.raises(ValueError):
        m.forward(fx['x'][:, :-1], fx['mels'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_device_loss_matches_the_reference(name):
    """wrnn_loss on the reference's own logits == the reference's loss value; on forward()'s logits == the reference's
    loss over the whole sequence."""
    from tacotronv2_wavernn_chinese_amd.losses import discretized_mix_logistic_loss, voc_loss
    fx = _load(name)
    m = _model(fx, 'batch')
    ysub = fx['y'][:, ::int(fx['sub_stride'])]
    got = float(voc_loss(m, fx['logits_sub'], ysub).item())
    assert abs(got - fx['loss_sub']) <= 2e-6 * max(1.0, abs(fx['loss_sub'])), (got, fx['loss_sub'])
    y_hat = m.forward(fx['x'], fx['mels'])
    full = float(voc_loss(m, y_hat, fx['y']).item())
    assert abs(full - fx['loss']) <= 2e-5 * max(1.0, abs(fx['loss'])), (full, fx['loss'])
    if fx['mode'] == 'MOL':
        assert float(discretized_mix_logistic_loss(m, y_hat, fx['y'][..., None]).item()) == full
    else:
        bad = fx['y'].copy()
        bad[0, 0] = 5000
        assert np.isnan(float(voc_loss(m, y_hat, bad).item()))     # torch raises for a target outside [0, n_classes)
