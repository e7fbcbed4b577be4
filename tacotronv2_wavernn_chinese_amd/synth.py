"""Synthetic WaveRNN weights / mels for tests and bench.py.

No trained checkpoint ships with the reference (``.MISSING_LARGE_BLOBS`` lists
``logs_wavernn/checkpoints/latest_weights.pyt``), so every parity and
performance run uses seeded synthetic weights.  The state_dict *layout* (key
names, shapes, dtypes) is the reference's on-disk contract:
``wavernn/models/fatchord_version.py:93-129`` (constructor) and
``:414-417`` (``load`` = ``torch.load`` of a flat state_dict).

The generator is numpy-only (PCG64) so the GPU box -- which has no
``/root/reference`` -- rebuilds bit-identical weights from the seed alone.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

DEFAULT_DIMS = dict(rnn_dims=512, fc_dims=512, bits=10, pad=2,
                    upsample_factors=(5, 5, 11), feat_dims=80,
                    compute_dims=128, res_out_dims=128, res_blocks=10,
                    hop_length=275, sample_rate=22050)


def n_classes_for(mode: str, bits: int) -> int:
    """fatchord_version.py:98-103."""
    if mode == 'RAW':
        return 2 ** bits
    if mode == 'MOL':
        return 30
    raise RuntimeError("Unknown model mode value - ", mode)


def make_state_dict(seed: int = 0, mode: str = 'RAW', variant: str = 'default',
                    **dims) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic state_dict with the reference's 148 keys.

    variant 'default': torch-like fan-in uniform init, randomised BatchNorm
    statistics (so the eval-mode BN path is exercised), perturbed up-conv taps
    (they are *learned* parameters, fatchord_version.py:77-78).
    variant 'peaky': same, with ``fc3.weight *= 128`` so the posterior is
    low-entropy like a trained model's (mean entropy ~2.7 nats of ln 1024 = 6.9;
    SURVEY.md section 7, hard part 5).
    """
    d = dict(DEFAULT_DIMS)
    d.update(dims)
    rnn, fc, feat = d['rnn_dims'], d['fc_dims'], d['feat_dims']
    comp, res_out, nblk = d['compute_dims'], d['res_out_dims'], d['res_blocks']
    pad = d['pad']
    aux = res_out // 4
    ncls = n_classes_for(mode, d['bits'])
    rng = np.random.Generator(np.random.PCG64(seed))

    def uni(shape, fan_in):
        k = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-k, k, size=shape).astype(np.float32)

    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd['step'] = np.zeros((1,), dtype=np.int64)

    def bn(prefix):
        sd[prefix + '.weight'] = rng.uniform(0.5, 1.5, size=(comp,)).astype(np.float32)
        sd[prefix + '.bias'] = (0.1 * rng.standard_normal(comp)).astype(np.float32)
        sd[prefix + '.running_mean'] = (0.1 * rng.standard_normal(comp)).astype(np.float32)
        sd[prefix + '.running_var'] = rng.uniform(0.5, 1.5, size=(comp,)).astype(np.float32)
        sd[prefix + '.num_batches_tracked'] = np.zeros((), dtype=np.int64)

    ksz = 2 * pad + 1
    sd['upsample.resnet.conv_in.weight'] = uni((comp, feat, ksz), feat * ksz)
    bn('upsample.resnet.batch_norm')
    for i in range(nblk):
        p = f'upsample.resnet.layers.{i}'
        sd[p + '.conv1.weight'] = uni((comp, comp, 1), comp)
        sd[p + '.conv2.weight'] = uni((comp, comp, 1), comp)
        bn(p + '.batch_norm1')
        bn(p + '.batch_norm2')
    sd['upsample.resnet.conv_out.weight'] = uni((res_out, comp, 1), comp)
    sd['upsample.resnet.conv_out.bias'] = uni((res_out,), comp)
    for li, s in enumerate(d['upsample_factors']):
        taps = 2 * s + 1
        w = (1.0 / taps) * (1.0 + 0.1 * rng.standard_normal(taps))
        sd[f'upsample.up_layers.{2 * li + 1}.weight'] = w.astype(np.float32).reshape(1, 1, 1, taps)

    sd['I.weight'] = uni((rnn, feat + aux + 1), feat + aux + 1)
    sd['I.bias'] = uni((rnn,), feat + aux + 1)
    sd['rnn1.weight_ih_l0'] = uni((3 * rnn, rnn), rnn)
    sd['rnn1.weight_hh_l0'] = uni((3 * rnn, rnn), rnn)
    sd['rnn1.bias_ih_l0'] = uni((3 * rnn,), rnn)
    sd['rnn1.bias_hh_l0'] = uni((3 * rnn,), rnn)
    sd['rnn2.weight_ih_l0'] = uni((3 * rnn, rnn + aux), rnn)
    sd['rnn2.weight_hh_l0'] = uni((3 * rnn, rnn), rnn)
    sd['rnn2.bias_ih_l0'] = uni((3 * rnn,), rnn)
    sd['rnn2.bias_hh_l0'] = uni((3 * rnn,), rnn)
    sd['fc1.weight'] = uni((fc, rnn + aux), rnn + aux)
    sd['fc1.bias'] = uni((fc,), rnn + aux)
    sd['fc2.weight'] = uni((fc, fc + aux), fc + aux)
    sd['fc2.bias'] = uni((fc,), fc + aux)
    sd['fc3.weight'] = uni((ncls, fc), fc)
    sd['fc3.bias'] = uni((ncls,), fc)
    if variant == 'peaky':
        sd['fc3.weight'] = (sd['fc3.weight'] * 128.0).astype(np.float32)
    elif variant != 'default':
        raise ValueError(f'unknown variant {variant!r}')
    return sd


def make_mels(seed: int, batch: int, frames: int, feat_dims: int = 80) -> np.ndarray:
    """(B, 80, T) float32 in [0,1) -- satisfies wavernn_gen.py:25-28."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((batch, feat_dims, frames), dtype=np.float32)


def make_dm_state_dict(seed: int = 0, hidden_size: int = 896, quantisation: int = 256,
                       out_scale: float = 16.0) -> "OrderedDict[str, np.ndarray]":
    """Seeded synthetic state_dict of the unconditioned dual-softmax model
    (``wavernn/models/deepmind_version.py:10-31``).  ``out_scale`` sharpens the two softmaxes (O2/O4) and the gate
    biases are randomised (the reference initialises them to zero) so every term of the step is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, S, Q = hidden_size, hidden_size // 2, quantisation

    def uni(shape, fan_in):
        k = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-k, k, size=shape).astype(np.float32)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd['bias_u'] = (0.1 * rng.standard_normal(H)).astype(np.float32)
    sd['bias_r'] = (0.1 * rng.standard_normal(H)).astype(np.float32)
    sd['bias_e'] = (0.1 * rng.standard_normal(H)).astype(np.float32)
    sd['R.weight'] = uni((3 * H, H), H)
    sd['O1.weight'] = uni((S, S), S); sd['O1.bias'] = uni((S,), S)
    sd['O2.weight'] = (uni((Q, S), S) * out_scale).astype(np.float32); sd['O2.bias'] = uni((Q,), S)
    sd['O3.weight'] = uni((S, S), S); sd['O3.bias'] = uni((S,), S)
    sd['O4.weight'] = (uni((Q, S), S) * out_scale).astype(np.float32); sd['O4.bias'] = uni((Q,), S)
    sd['I_coarse.weight'] = uni((3 * S, 2), 2)
    sd['I_fine.weight'] = uni((3 * S, 3), 3)
    return sd
