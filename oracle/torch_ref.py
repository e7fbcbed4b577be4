"""TEST INFRASTRUCTURE -- not product code.

Restatement of the reference's TRAINING computation in plain torch ops, driven by a flat state_dict, so that autograd
gives reference gradients on a box that has no /root/reference (the GPU box):
    WaveRNN.forward                      wavernn/models/fatchord_version.py:131-167
    UpsampleNetwork / MelResNet / ...    :13-89   (BatchNorm in training mode: batch statistics)
    the training script's loss           wavernn_train.py:82,112-121; wavernn/utils/distribution.py:16-84
Imported only by tests/.  Parity pin: tests/test_train_step.py::test_torch_restatement_equals_the_reference_module runs it
against the UNMODIFIED reference module (forward value, loss and every parameter gradient) where /root/reference exists, and
tests/golden/train_*.npz (minted from the reference by `python -m oracle.make_golden train`) pin it everywhere else.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, training):
    # nn.BatchNorm1d(dims), eps 1e-5: batch statistics when training (running-stat updates do not affect the output)
    return F.batch_norm(x, sd[prefix + '.running_mean'].clone(), sd[prefix + '.running_var'].clone(), sd[prefix + '.weight'],
                        sd[prefix + '.bias'], training=training, momentum=0.1, eps=1e-5)


def upsample(sd: Dict[str, torch.Tensor], mels: torch.Tensor, upsample_factors=(5, 5, 11), pad=2, training=True):
    """UpsampleNetwork.forward (:82-89): mels (B, n_mels, T + 2*pad) -> (mels_up (B, L, n_mels), aux (B, L, res_out))."""
    x = F.conv1d(mels, sd['upsample.resnet.conv_in.weight'])                    # MelResNet.forward :42-48
    x = F.relu(_bn(x, sd, 'upsample.resnet.batch_norm', training))
    i = 0
    while f'upsample.resnet.layers.{i}.conv1.weight' in sd:                     # ResBlock.forward :21-28
        p = f'upsample.resnet.layers.{i}'
        res = x
        x = F.relu(_bn(F.conv1d(x, sd[p + '.conv1.weight']), sd, p + '.batch_norm1', training))
        x = _bn(F.conv1d(x, sd[p + '.conv2.weight']), sd, p + '.batch_norm2', training)
        x = x + res
        i += 1
    aux = F.conv1d(x, sd['upsample.resnet.conv_out.weight'], sd['upsample.resnet.conv_out.bias'])
    hop = int(np.prod(upsample_factors))
    aux = aux.repeat_interleave(hop, dim=2)                                     # Stretch2d(total_scale, 1) :57-61
    m = mels.unsqueeze(1)
    for li, s in enumerate(upsample_factors):
        m = m.repeat_interleave(s, dim=3)
        m = F.conv2d(m, sd[f'upsample.up_layers.{2 * li + 1}.weight'], padding=(0, s))
    indent = pad * hop
    m = m.squeeze(1)[:, :, indent:-indent]
    return m.transpose(1, 2), aux.transpose(1, 2)


def _gru(x, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU(batch_first=True), one layer, h0 = 0 (:141-142): gate order r, z, n."""
    B, L, _ = x.shape
    H = w_hh.shape[1]
    gi = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    out = []
    for t in range(L):
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1.0 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=1)


def loop_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, mels_up: torch.Tensor, aux: torch.Tensor):
    """forward() from the upsampled conditioning on (:145-167): fc3 outputs (B, L, n_classes)."""
    A = aux.shape[2] // 4
    a1, a2, a3, a4 = (aux[:, :, i * A:(i + 1) * A] for i in range(4))
    h = torch.cat([x.unsqueeze(-1), mels_up, a1], dim=2) @ sd['I.weight'].t() + sd['I.bias']
    res = h
    h = _gru(h, sd['rnn1.weight_ih_l0'], sd['rnn1.weight_hh_l0'], sd['rnn1.bias_ih_l0'], sd['rnn1.bias_hh_l0']) + res
    res = h
    h = _gru(torch.cat([h, a2], dim=2), sd['rnn2.weight_ih_l0'], sd['rnn2.weight_hh_l0'], sd['rnn2.bias_ih_l0'], sd['rnn2.bias_hh_l0']) + res
    h = F.relu(torch.cat([h, a3], dim=2) @ sd['fc1.weight'].t() + sd['fc1.bias'])
    h = F.relu(torch.cat([h, a4], dim=2) @ sd['fc2.weight'].t() + sd['fc2.bias'])
    return h @ sd['fc3.weight'].t() + sd['fc3.bias']


def _log_sum_exp(x):                                                            # distribution.py:6-12
    m, _ = torch.max(x, dim=-1)
    m2, _ = torch.max(x, dim=-1, keepdim=True)
    return m + torch.log(torch.sum(torch.exp(x - m2), dim=-1))


def discretized_mix_logistic_loss(y_hat, y, num_classes=65536, log_scale_min=None):
    """distribution.py:16-84 with reduce=True; y_hat (B, L, 30) as forward() returns it, y (B, L)."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))
    y_hat = y_hat.permute(0, 2, 1)                                              # :20 (B, C, T)
    nr_mix = y_hat.size(1) // 3
    y_hat = y_hat.transpose(1, 2)                                               # :27 (B, T, C)
    logit_probs = y_hat[:, :, :nr_mix]
    means = y_hat[:, :, nr_mix:2 * nr_mix]
    log_scales = torch.clamp(y_hat[:, :, 2 * nr_mix:3 * nr_mix], min=log_scale_min)
    y = y.unsqueeze(-1).expand_as(means)
    centered_y = y - means
    inv_stdv = torch.exp(-log_scales)
    plus_in = inv_stdv * (centered_y + 1. / (num_classes - 1))
    cdf_plus = torch.sigmoid(plus_in)
    min_in = inv_stdv * (centered_y - 1. / (num_classes - 1))
    cdf_min = torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_stdv * centered_y
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    inner_inner_cond = (cdf_delta > 1e-5).float()
    inner_inner_out = inner_inner_cond * torch.log(torch.clamp(cdf_delta, min=1e-12)) + \
        (1. - inner_inner_cond) * (log_pdf_mid - np.log((num_classes - 1) / 2))
    inner_cond = (y > 0.999).float()
    inner_out = inner_cond * log_one_minus_cdf_min + (1. - inner_cond) * inner_inner_out
    cond = (y < -0.999).float()
    log_probs = cond * log_cdf_plus + (1. - cond) * inner_out
    log_probs = log_probs + F.log_softmax(logit_probs, -1)
    return -torch.mean(_log_sum_exp(log_probs))


def loss_of(mode: str, y_hat: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """wavernn_train.py:112-121."""
    if mode == 'RAW':
        return F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), y.long().unsqueeze(-1))
    return discretized_mix_logistic_loss(y_hat, y.float())


def training_step(state_dict: Dict[str, np.ndarray], mode: str, x: np.ndarray, mels: np.ndarray, y: np.ndarray, dtype=torch.float32,
                  device='cpu', training=True) -> dict:
    """loss + gradients of every floating-point parameter for one (x, mels, y) batch, as `loss.backward()` on the reference
    module in train() mode gives them.  Returns dict(loss, logits, grads {key: ndarray}, d_mels_up, d_aux)."""
    sd = {}
    for k, v in state_dict.items():
        t = torch.as_tensor(np.asarray(v))
        if t.is_floating_point():
            t = t.to(device=device, dtype=dtype).clone().requires_grad_(not k.endswith(('running_mean', 'running_var')))
        sd[k] = t
    xt = torch.as_tensor(x).to(device=device, dtype=dtype)
    mt = torch.as_tensor(mels).to(device=device, dtype=dtype)
    yt = torch.as_tensor(y).to(device=device)
    mels_up, aux = upsample(sd, mt, training=training)
    mels_up.retain_grad()
    aux.retain_grad()
    y_hat = loop_forward(sd, xt, mels_up, aux)
    loss = loss_of(mode, y_hat, yt)
    loss.backward()
    grads = {k: v.grad.detach().cpu().numpy() for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    return dict(loss=float(loss.detach()), logits=y_hat.detach().cpu().numpy(), grads=grads, d_mels_up=mels_up.grad.detach().cpu().numpy(),
                d_aux=aux.grad.detach().cpu().numpy(), mels_up=mels_up.detach().cpu().numpy(), aux=aux.detach().cpu().numpy())


def float64_logits_at(state_dict: Dict[str, np.ndarray], mel: np.ndarray, x_fed: np.ndarray, steps, pad: int = 2, device='cuda') -> np.ndarray:
    """fc3 outputs in FLOAT64 at `steps` of ONE row whose loop was fed the values `x_fed` (L,): generate()'s arithmetic (:183-223: zero-padded
    mel, eval-mode upsample network, x_0 = 0 :197, then the value appended to `output` :227,:236) with every operation in double precision.
    Used by the parity tests to say which side a near-tie of the sampler's race falls on in exact arithmetic.  mel (n_mels, T) float32."""
    steps = [int(t) for t in steps]
    n = max(steps) + 1
    with torch.no_grad():
        sd = {k: torch.as_tensor(np.asarray(v)).to(device=device, dtype=torch.float64) for k, v in state_dict.items()
              if np.asarray(v).dtype.kind == 'f'}
        m = torch.as_tensor(np.asarray(mel)).to(device=device, dtype=torch.float64)[None]
        m = F.pad(m, (pad, pad))                                                    # pad_tensor(side='both') :183, :281-291
        mels_up, aux = upsample(sd, m, pad=pad, training=False)
        x = torch.zeros(1, n, dtype=torch.float64, device=device)
        x[0, 1:] = torch.as_tensor(np.asarray(x_fed[:n - 1])).to(device=device, dtype=torch.float64)
        y = loop_forward(sd, x, mels_up[:, :n], aux[:, :n])
        return y[0, steps].cpu().numpy()
