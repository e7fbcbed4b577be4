"""Drop-in host side of the WaveRNN vocoder (mel -> wav) for MI355X.

Mirrors the call surface of ``wavernn/models/fatchord_version.py`` of the
reference: ``WaveRNN(rnn_dims, fc_dims, bits, pad, upsample_factors, feat_dims,
compute_dims, res_out_dims, res_blocks, hop_length, sample_rate, mode)``
(:93-95), ``generate(mels, save_path, batched, target, overlap, mu_law)``
(:169-264), ``load``/``save``/``get_step``/``num_params`` (:407-429), the flat
``state_dict`` key layout, and the reference's observable quirks (listed in
``generate``).  The arithmetic of the hot path runs in libwavernn_amd.so (HIP,
gfx950) through the C-ABI of ``include/wavernn_amd.h``; this module only holds
parameters, moves pointers, and performs the float64 epilogue (:243-260).

The class is an ``nn.Module`` purely as a parameter container: sub-module
names and construction order follow the reference so that ``state_dict()``
keys, ``load_state_dict`` and the default initialisation under a given
``torch.manual_seed`` are interchangeable with it.  ``forward`` (teacher-forced
pass, :131-167) runs the same loop kernels with the fed-back value forced;
the losses of the training script live in ``losses.py``.
"""
from __future__ import annotations

import math
import sys
import time
import warnings
from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _cabi
from .dsp import decode_mu_law, save_wav


def _bag(**mods) -> nn.Module:
    m = nn.Module()
    for k, v in mods.items():
        m.add_module(k, v)
    return m


def _make_upsample(feat_dims, upsample_scales, compute_dims, res_blocks, res_out_dims, pad) -> nn.Module:
    """Parameter containers with the key layout of UpsampleNetwork/MelResNet (:31-80)."""
    resnet = nn.Module()
    resnet.add_module('conv_in', nn.Conv1d(feat_dims, compute_dims, kernel_size=2 * pad + 1, bias=False))
    resnet.add_module('batch_norm', nn.BatchNorm1d(compute_dims))
    blocks = nn.ModuleList()
    for _ in range(res_blocks):
        blocks.append(_bag(conv1=nn.Conv1d(compute_dims, compute_dims, kernel_size=1, bias=False),
                           conv2=nn.Conv1d(compute_dims, compute_dims, kernel_size=1, bias=False),
                           batch_norm1=nn.BatchNorm1d(compute_dims),
                           batch_norm2=nn.BatchNorm1d(compute_dims)))
    resnet.add_module('layers', blocks)
    resnet.add_module('conv_out', nn.Conv1d(compute_dims, res_out_dims, kernel_size=1))
    up = nn.Module()
    up.add_module('resnet', resnet)
    layers = nn.ModuleList()
    for s in upsample_scales:
        layers.append(nn.Identity())  # slot of the parameter-free Stretch2d (:74)
        conv = nn.Conv2d(1, 1, kernel_size=(1, 2 * s + 1), padding=(0, s), bias=False)
        conv.weight.data.fill_(1. / (2 * s + 1))  # :78
        layers.append(conv)
    up.add_module('up_layers', layers)
    return up


def fold_target(total_len: int, overlap: int, n_folds: int) -> int:
    """Smallest ``target`` for which ``fold_with_overlap`` (:293-340) cuts ``total_len`` samples into at most
    ``n_folds`` folds: ceil((total_len - overlap) / n_folds) - overlap, never below ``overlap``."""
    return max(-(-(int(total_len) - int(overlap)) // int(n_folds)) - int(overlap), int(overlap), 1)


def fold_count(total_len: int, target: int, overlap: int) -> int:
    """Number of folds ``fold_with_overlap`` cuts ``total_len`` samples into (:319-325; = ``wrnn_plan``'s rows)."""
    num_folds, remaining = divmod(int(total_len) - int(overlap), int(target) + int(overlap))
    return num_folds + (1 if remaining != 0 else 0)


# Microseconds per lock-step loop step of ONE XCD team on an MI355X, by what the team runs (profiles/r06_fold_latency_{raw,mol}.txt):
# 'team2' the latency kernel (one row per team), 'cs4' / 'cs8' the batch kernel at <= 4 / <= 8 rows per team.  Only the RATIOS matter to
# ``fold_plan`` (it compares fold counts on one device); a device with another clock shifts all three alike.
STEP_US = {'RAW': {'team2': 3.30, 'cs4': 5.28, 'cs8': 7.52}, 'MOL': {'team2': 3.43, 'cs4': 4.85, 'cs8': 8.16}}
ROWS_PER_TEAM_MAX = 8   # WRNN_BATCH_MAX_ROWS


def predicted_loop_us(rows: int, steps: int, n_teams: int, mode: str = 'RAW') -> float:
    """Loop time ``WRNN_KERNEL_AUTO`` needs for ``rows`` rows of ``steps`` steps on ``n_teams`` XCD teams (api.hip: rows <= teams run one per
    team on the latency kernel; more rows are spread evenly, ceil(rows / teams) <= 8 per team batch in lock-step, a team runs its batches
    back to back)."""
    us = STEP_US[mode]
    if rows <= n_teams:
        return steps * us['team2']
    rpb = min(-(-rows // n_teams), ROWS_PER_TEAM_MAX)
    passes = -(-(-(-rows // rpb)) // n_teams)
    return steps * passes * (us['cs4'] if rpb <= 4 else us['cs8'])


def fold_plan(total_len: int, overlap: int, n_teams: int = 8, mode: str = 'RAW', min_target: int = 0):
    """The fold count that minimises the predicted loop time of ONE utterance in the reference's batched mode (``target='auto'``):
    every count from 1 to 2 x 8 x teams is priced as steps(target, overlap) x us/step(rows per team) x passes -- fewer, longer folds
    cost steps, more folds cost the 2 x overlap samples each one generates twice and, past 8 per team, a second pass.  On an MI355X
    a 5 s clip lands at 64 folds (8 per team, 2 265 steps, ~17 ms) against 47 ms for one fold per team.  Returns
    (target, folds, predicted_us); ties go to the smaller fold count (fewer crossfades).  ``min_target`` bounds the crossfade DENSITY
    from below (the latency optimum of a 5 s clip is a 550-sample crossfade every 1 715 samples: a third of the audio lies inside one;
    ``min_target=5500`` gives 18 folds / 35 ms instead of 64 / 17 ms); a one-fold plan is always admissible."""
    best = None
    for n in range(1, 2 * ROWS_PER_TEAM_MAX * max(n_teams, 1) + 1):
        target = fold_target(total_len, overlap, n)
        rows = fold_count(total_len, target, overlap)
        if rows < 1 or (target < min_target and n > 1):
            continue
        cost = predicted_loop_us(rows, target + 2 * overlap, max(n_teams, 1), mode)
        if best is None or cost < best[2] * (1.0 - 1e-9):
            best = (target, rows, cost)
    if best is None:
        raise ValueError(f'no fold plan for {total_len} samples with overlap {overlap}')
    return best


_CU_COUNT = {}   # torch.cuda.get_device_properties costs ~0.1 s on its first call (amdsmi): asked once per device


def _device_teams(dev) -> int:
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _CU_COUNT:
        _CU_COUNT[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return max(1, min(8, _CU_COUNT[idx] // 32))


def reference_noise(mode: str, rows: int, steps: int, n_classes: int, rnn_dims: int, aux_dims: int, device, chunk_bytes: int = 64 << 20):
    """The sampling noise of the reference's ``generate`` (:169-264) drawn from the GLOBAL torch CPU generator exactly as the reference
    consumes it, so that ``torch.manual_seed(s); model.generate(..., noise_mode='reference')`` is the reference's own output for seed s
    (model on the CPU there; a reference model on a CUDA device draws from that device's generator, which cannot be replayed):
      1. ``get_gru_cell`` builds two ``nn.GRUCell`` objects whose default initialisation draws from the generator (:178-179, :273-279);
      2. RAW: ``Categorical.sample()`` = ``torch.multinomial(p, 1, True)`` = argmax p / q with ``q = empty(rows, n_classes).exponential_(1)``,
         one call per step (:231-235);  MOL: ``uniform_(1e-5, 1 - 1e-5)`` on (1, rows, 10) and then on (1, rows), per step
         (``wavernn/utils/distribution.py:106,118``).
    One generator call per step like the reference (``exponential_`` seeds a per-call stream whose content depends on the call's size),
    written straight into a host chunk and uploaded.  Returns (noise1, noise2) device tensors laid out as ``wrnn_sample_opts`` wants them:
    RAW (steps, rows, n_classes), None;  MOL (steps, rows, 10), (steps, rows)."""
    device = torch.device(device)
    pin = device.type == 'cuda'
    nn.GRUCell(rnn_dims, rnn_dims)                 # get_gru_cell(self.rnn1): draws discarded, as in the reference
    nn.GRUCell(rnn_dims + aux_dims, rnn_dims)      # get_gru_cell(self.rnn2)
    width = n_classes if mode == 'RAW' else 11
    chunk = int(max(1, min(steps, chunk_bytes // (rows * width * 4))))
    if mode == 'RAW':
        out1, out2 = torch.empty((steps, rows, n_classes), dtype=torch.float32, device=device), None
        b1 = torch.empty((chunk, rows, n_classes), dtype=torch.float32, pin_memory=pin)
        for t0 in range(0, steps, chunk):
            n = min(chunk, steps - t0)
            for t in range(n):
                b1[t].exponential_(1)
            out1[t0:t0 + n].copy_(b1[:n])
        # exponential_ may return 0 (p / q = inf in the reference: that class wins); the log-domain race needs a finite -log q
        out1.clamp_(min=1.2e-38)
    else:
        out1 = torch.empty((steps, rows, 10), dtype=torch.float32, device=device)
        out2 = torch.empty((steps, rows), dtype=torch.float32, device=device)
        b1 = torch.empty((chunk, 1, rows, 10), dtype=torch.float32, pin_memory=pin)
        b2 = torch.empty((chunk, 1, rows), dtype=torch.float32, pin_memory=pin)
        for t0 in range(0, steps, chunk):
            n = min(chunk, steps - t0)
            for t in range(n):
                b1[t].uniform_(1e-5, 1.0 - 1e-5)
                b2[t].uniform_(1e-5, 1.0 - 1e-5)
            out1[t0:t0 + n].copy_(b1[:n, 0])
            out2[t0:t0 + n].copy_(b2[:n, 0])
    return out1, out2


_NOISE_REFERENCE = -1   # host-side mode: resolved to NOISE_INJECTED with the reference's own draws before the C-ABI call
_NOISE_MODES = {'philox': _cabi.NOISE_PHILOX, 'injected': _cabi.NOISE_INJECTED, 'argmax': _cabi.NOISE_ARGMAX, 'reference': _NOISE_REFERENCE}


class WaveRNN(nn.Module):
    def __init__(self, rnn_dims, fc_dims, bits, pad, upsample_factors,
                 feat_dims, compute_dims, res_out_dims, res_blocks,
                 hop_length, sample_rate, mode='RAW'):
        super().__init__()
        self.mode = mode
        self.pad = pad
        if self.mode == 'RAW':
            self.n_classes = 2 ** bits
        elif self.mode == 'MOL':
            self.n_classes = 30
        else:
            # the reference builds the exception without raising it (:102-103) and then
            # fails on the missing n_classes; raise the same type explicitly
            raise RuntimeError("Unknown model mode value - ", self.mode)
        self.bits = bits
        self.rnn_dims = rnn_dims
        self.fc_dims = fc_dims
        self.aux_dims = res_out_dims // 4
        self.hop_length = hop_length
        self.sample_rate = sample_rate
        self.feat_dims = feat_dims
        self._ctor = dict(rnn_dims=rnn_dims, fc_dims=fc_dims, bits=bits, pad=pad,
                          upsample_factors=tuple(int(s) for s in upsample_factors), feat_dims=feat_dims,
                          compute_dims=compute_dims, res_out_dims=res_out_dims, res_blocks=res_blocks,
                          hop_length=hop_length, sample_rate=sample_rate, mode=mode)

        self.upsample = _make_upsample(feat_dims, upsample_factors, compute_dims, res_blocks, res_out_dims, pad)
        self.I = nn.Linear(feat_dims + self.aux_dims + 1, rnn_dims)
        self.rnn1 = nn.GRU(rnn_dims, rnn_dims, batch_first=True)
        self.rnn2 = nn.GRU(rnn_dims + self.aux_dims, rnn_dims, batch_first=True)
        self.fc1 = nn.Linear(rnn_dims + self.aux_dims, fc_dims)
        self.fc2 = nn.Linear(fc_dims + self.aux_dims, fc_dims)
        self.fc3 = nn.Linear(fc_dims, self.n_classes)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.num_params()

        # native state (created lazily, per device)
        self._native: Optional[_cabi.NativeVocoder] = None
        self._native_key = None
        # knobs that are not part of the reference signature
        self.kernel = _cabi.KERNEL_AUTO
        # True: every training call (training_loss, forward / backward in train() mode) waits for its kernels and raises on a device-side
        # error (a busy GPU, a timed-out team kernel) -- safe, but the host cannot queue the rest of the iteration (the upsample network's
        # backward, clipping, Adam: ~150 small launches) under the running step.  'deferred': no wait; the caller asks once per iteration
        # with ``training_status()`` (``train.voc_train_loop`` does, at its ``loss.item()``, and before it writes a checkpoint).
        self.check_device_errors = True
        self.fold_min_target = 0          # target='auto': never plan folds shorter than this many samples (0: latency optimum, see fold_plan)
        self.busy_retry_seconds = 0.05   # WRNN_ERR_BUSY under AUTO: wait this long, retry the team kernel once, then fall back (loudly)
        self.verbose = True
        self.last_timing: Optional[dict] = None

    # ------------------------------------------------------------------ native
    def _device_index(self) -> int:
        dev = next(self.parameters()).device
        if dev.type == 'cuda':
            return dev.index if dev.index is not None else torch.cuda.current_device()
        if not torch.cuda.is_available():
            raise RuntimeError('WaveRNN.generate needs an MI355X (HIP) device: no GPU is visible and there is '
                               'no CPU fallback in this package')
        return torch.cuda.current_device()

    def _weights_key(self, dev: int):
        """Cheap fingerprint of the parameters the native handle was packed from: identity, in-place version counter
        and storage address of every tensor.  Catches ``load_state_dict``, optimizer steps, ``p.data = t`` (the idiom of
        the reference's ``get_gru_cell``) and ``.to()``.  Writes through ``p.data`` that keep the storage
        (``p.data.copy_()``, ``p.data.fill_()``) bump neither: call :meth:`invalidate_native` after those."""
        # `step` and BatchNorm's `num_batches_tracked` are never read by wrnn_load_weights: leaving them out keeps forward()
        # (which bumps `step` in place, :139) from repacking and re-uploading every weight on every call
        # (read from the sub-modules' own `_parameters` / `_buffers` dicts: `parameters()` + `named_buffers()` with their de-duplication
        # sets cost 0.5 ms per call, 3 % of a 19 ms single-utterance generate)
        objs = []
        for m in self.modules():
            objs.extend(m._parameters.values())
            objs.extend(b for n, b in m._buffers.items() if n != 'step' and n != 'num_batches_tracked')
        return (dev,) + tuple((id(p), p._version, p.data_ptr()) for p in objs if p is not None)

    def invalidate_native(self):
        """Force the next ``generate`` to repack the weights into the native handle."""
        self._native_key = None

    def _native_handle(self) -> _cabi.NativeVocoder:
        """The wrnn_handle for the current device WITHOUT (re)packing the inference weights: what ``wrnn_train_step`` needs
        (dims, mode, workspace) -- the parameters change every optimizer step and are read where torch keeps them."""
        dev = self._device_index()
        if self._native is None or self._native.device != dev:
            if self._native is not None:
                self._native.close()
            self._native = _cabi.NativeVocoder(device=dev, **self._ctor)
            self._native_key = None
        return self._native

    def native(self) -> _cabi.NativeVocoder:
        """The wrnn_handle for the current device, repacked if parameters changed."""
        dev = self._device_index()
        key = self._weights_key(dev)
        if self._native is None or self._native.device != dev:
            if self._native is not None:
                self._native.close()
            self._native = _cabi.NativeVocoder(device=dev, **self._ctor)
            self._native_key = None
        if self._native_key != key:
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items()}
            self._native.load_weights(sd)
            self._native_key = key
        return self._native

    def forward(self, x, mels):
        """``forward`` of the reference (:131-167).

        * ``self.training`` and gradients enabled (what ``y_hat = model(x, m)`` is in the reference's training loop,
          ``wavernn_train.py:103-110``): the DIFFERENTIABLE pass -- BatchNorm on batch statistics like the reference module in
          train() mode, loop layers in ``wrnn_train_forward``; the returned y_hat (B, L, n_classes) carries an autograd graph whose
          backward is ``wrnn_train_backward``, so ``loss_func(y_hat, y).backward()`` with ANY torch loss fills every ``.grad``: the
          reference's training script runs unchanged (``training_loss`` fuses the script's own loss and its gradient into one call).
        * otherwise (``eval()`` or ``torch.no_grad()``): the teacher-forced pass on the hot path's own loop kernels, see below.

        Teacher-forced pass on the loop kernels: ``x`` (B, L) is the
        input sample sequence, ``mels`` (B, n_mels, T + 2*pad) the mel window already padded with ``pad`` context frames on
        both sides (what the training collate hands over, :143), L = T * hop.  Returns the fc3 outputs (B, L, n_classes) --
        logits (RAW) / mixture parameters (MOL) -- as a float32 tensor on the model's device.  Inference only (no autograd
        graph): it is the loop of ``generate`` with the fed-back value forced to ``x`` and the sampler's result ignored.
        Increments ``step`` like the reference (:139).
        """
        x_t = torch.as_tensor(x, dtype=torch.float32)
        mels_t = torch.as_tensor(mels)
        if mels_t.dim() != 3 or x_t.dim() != 2:
            raise ValueError(f'expected x (B, L) and mels (B, n_mels, T + 2*pad), got {tuple(x_t.shape)}, {tuple(mels_t.shape)}')
        T = mels_t.size(-1) - 2 * self.pad
        if T < 1 or x_t.shape != (mels_t.size(0), T * self.hop_length):
            raise ValueError(f'x must be (B, {max(T, 0) * self.hop_length}) for mels {tuple(mels_t.shape)} (pad {self.pad}, hop {self.hop_length})')
        self.step += 1
        if self.training and torch.is_grad_enabled():
            dev = next(self.parameters()).device
            if dev.type != 'cuda':
                raise RuntimeError('forward() in training mode needs the model on the MI355X (no CPU path in this package)')
            mels_up, aux = self.upsample_torch(mels_t.to(device=dev, dtype=torch.float32))
            return _LoopForwardFn.apply(self, x_t.to(device=dev).contiguous(), mels_up.contiguous(), aux.contiguous(), *self._loop_params())
        xs = x_t.detach().cpu().numpy()
        x_forced = np.zeros((xs.shape[1], xs.shape[0]), np.float32)
        x_forced[:-1] = xs[:, 1:].T                      # the value fed to step t + 1 is x[:, t + 1]
        kw = dict(noise_mode=_cabi.NOISE_ARGMAX) if self.mode == 'RAW' else \
            dict(noise_mode=_cabi.NOISE_PHILOX, seed=0)  # the drawn samples are discarded
        res = self.generate_raw(mels_t, False, 11000, 550, x_forced=x_forced, x_init=xs[:, 0], want_logits=True,
                                mels_padded=True, **kw)
        return res['logits'].permute(1, 0, 2).contiguous()

    # ---------------------------------------------------------------- training
    def upsample_torch(self, mels):
        """``self.upsample(mels)`` of the reference (UpsampleNetwork.forward :82-89, MelResNet :42-48, ResBlock :21-28,
        Stretch2d :57-61) on the framework's own ops, for TRAINING: BatchNorm follows ``self.training`` (batch statistics +
        running-stat updates in train mode) and the result carries an autograd graph, so the upsample network's parameters
        train through ``loss.backward()``.  mels (B, n_mels, T + 2*pad) -> (mels_up (B, L, n_mels), aux (B, L, res_out))."""
        r = self.upsample.resnet
        x = F.relu(r.batch_norm(r.conv_in(mels)))
        for blk in r.layers:
            res = x
            x = F.relu(blk.batch_norm1(blk.conv1(x)))
            x = blk.batch_norm2(blk.conv2(x))
            x = x + res
        aux = r.conv_out(x)                                                   # (B, res_out, T)
        aux = aux.repeat_interleave(self.hop_length, dim=2)                   # Stretch2d(total_scale, 1) (:70,:84)
        m = mels.unsqueeze(1)
        for i, s in enumerate(self._ctor['upsample_factors']):
            m = m.repeat_interleave(s, dim=3)                                 # Stretch2d(s, 1)
            m = self.upsample.up_layers[2 * i + 1](m)
        indent = self.pad * self.hop_length
        m = m.squeeze(1)[:, :, indent:-indent]
        return m.transpose(1, 2), aux.transpose(1, 2)

    def _loop_params(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in _cabi.LOOP_PARAM_KEYS]

    def training_loss(self, x, mels, y, return_logits=False):
        """One training step's forward as the reference's loop body computes it (``wavernn_train.py:103-121``):
        ``y_hat = model(x, mels)`` + ``F.cross_entropy`` (RAW) / ``discretized_mix_logistic_loss`` (MOL), as a 0-dim tensor WITH
        an autograd graph: ``loss.backward()`` fills ``.grad`` of every parameter, after which ``clip_grad_norm_`` /
        ``optimizer.step()`` work as in the reference.  The loop layers (I, rnn1, rnn2, fc1-3: 97 % of the FLOPs) run
        forward AND backward in ``wrnn_train_step`` (csrc/train.hip: batched fp32 MFMA GEMMs over all (batch, step) pairs +
        BPTT step kernels replayed from hipGraphs); the upsample network runs on the framework's ops (``upsample_torch``).
        x (B, L) float inputs, mels (B, n_mels, T + 2*pad), y (B, L) int labels (RAW) / float targets (MOL), L = T * hop.
        Increments ``step`` like ``forward`` (:139)."""
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('training_loss needs the model on the MI355X (no CPU path in this package)')
        x = torch.as_tensor(x).to(device=dev, dtype=torch.float32).contiguous()
        mels = torch.as_tensor(mels).to(device=dev, dtype=torch.float32)
        y = torch.as_tensor(y).to(device=dev)
        y = (y.to(torch.int32) if self.mode == 'RAW' else y.to(torch.float32)).contiguous()
        T = mels.size(-1) - 2 * self.pad
        if mels.dim() != 3 or T < 1 or tuple(x.shape) != (mels.size(0), T * self.hop_length) or tuple(y.shape) != tuple(x.shape):
            raise ValueError(f'expected x, y (B, {max(T, 0) * self.hop_length}) for mels {tuple(mels.shape)}')
        self.step += 1
        mels_up, aux = self.upsample_torch(mels)
        if not torch.is_grad_enabled():   # a validation pass: forward + loss only (wrnn_train_step without gradient outputs)
            mels_up, aux = mels_up.contiguous(), aux.contiguous()
            nat = self._native_handle()
            loss = torch.empty((), dtype=torch.float32, device=dev)
            logits = torch.empty((x.size(0), x.size(1), self.n_classes), dtype=torch.float32, device=dev) if return_logits else None
            ps = [p.detach().contiguous() for p in self._loop_params()]
            with torch.cuda.device(dev):
                st = torch.cuda.current_stream(dev).cuda_stream
                nat.train_step([p.data_ptr() for p in ps], None, x.data_ptr(), mels_up.data_ptr(), aux.data_ptr(), y.data_ptr(), x.size(0), x.size(1),
                               loss.data_ptr(), logits.data_ptr() if return_logits else 0, 0, 0, st)
                if self.check_device_errors is True:
                    nat.sync_status(st)
            return (loss, logits) if return_logits else loss
        out = _LoopTrainFn.apply(self, x, y, return_logits, mels_up.contiguous(), aux.contiguous(), *self._loop_params())
        return out if return_logits else out[0]

    def training_status(self):
        """Waits for the model's stream and raises ``WrnnError`` if a training kernel launched since the last forward pass reported a
        device-side error (``wrnn_sync_status``).  For ``check_device_errors = 'deferred'``: call it once per iteration, before the weights
        of that iteration are trusted (written to a checkpoint)."""
        dev = next(self.parameters()).device
        if dev.type != 'cuda' or self._native is None:
            return
        with torch.cuda.device(dev):
            self._native.sync_status(torch.cuda.current_stream(dev).cuda_stream)

    # ---------------------------------------------------------------- generate
    def generate_raw(self, mels, batched, target, overlap, *, noise_mode=_cabi.NOISE_PHILOX, seed=0,
                     noise1=None, noise2=None, x_forced=None, want_logits=False, kernel=None, x_init=None,
                     mels_padded=False, frames=None, batch_rows=0, team2_segment=0):
        """Device part of generate() (:183-241).  Returns dict(samples (rows, L) float32 cuda tensor,
        labels (rows, L) int32 cuda tensor, logits or None, rows, steps).

        noise_mode: ``_cabi.NOISE_PHILOX`` / ``'philox'`` (device counter RNG keyed by ``seed``), ``NOISE_INJECTED`` / ``'injected'``
        (noise1 / noise2 given), ``NOISE_ARGMAX`` / ``'argmax'`` (RAW, greedy), or ``'reference'``: the draws the reference's own
        ``generate`` makes from the global torch CPU generator in its own order (``reference_noise``), injected -- under
        ``torch.manual_seed(s)`` the call then reproduces the reference's output for that seed.
        noise1/noise2/x_forced: array-likes laid out like the reference consumes them (step-major):
        RAW noise1 (L, rows, n_classes) Exp(1) draws; MOL noise1 (L, rows, 10), noise2 (L, rows).
        frames: optional (B,) valid frames per utterance of a ragged, right-zero-padded batch (``wrnn_sample_opts.frames_dev``):
        row b runs frames[b] * hop steps, the rest of its output row is left unwritten (here: zero).
        batch_rows / team2_segment: tuning knobs of the BATCH / TEAM2 kernels (0 = the library's choice).
        """
        if isinstance(noise_mode, str):
            if noise_mode not in _NOISE_MODES:
                raise ValueError(f'noise_mode must be one of {sorted(_NOISE_MODES)} or a WRNN_NOISE_* id, got {noise_mode!r}')
            noise_mode = _NOISE_MODES[noise_mode]
        nat = self.native()
        dev = torch.device('cuda', nat.device)
        with torch.cuda.device(dev):
            mels_t = torch.as_tensor(mels).to(device=dev, dtype=torch.float32).contiguous()
            if mels_t.dim() != 3:
                raise ValueError(f'expected mels shaped (B, n_mels, T), got {tuple(mels_t.shape)}')
            B, F, T = mels_t.shape
            if mels_padded:
                T -= 2 * self.pad                        # forward()'s layout: `pad` context frames on both sides
            if F != self.feat_dims:   # the reference dies in conv_in with a channel mismatch (:43); a (T, n_mels) array lands here too
                raise ValueError(f'expected mels shaped (B, {self.feat_dims}, T), got {tuple(mels_t.shape)}')
            rows, steps = nat.plan(B, T, batched, target, overlap)
            ragged = frames is not None
            samples = (torch.zeros if ragged else torch.empty)((rows, steps), dtype=torch.float32, device=dev)
            labels = (torch.zeros if ragged else torch.empty)((rows, steps), dtype=torch.int32, device=dev)
            keep = []
            fr = 0
            if ragged:
                fr_t = torch.as_tensor(frames).to(device=dev, dtype=torch.int32).contiguous()
                if tuple(fr_t.shape) != (B,):
                    raise ValueError(f'frames must have shape ({B},), got {tuple(fr_t.shape)}')
                keep.append(fr_t)
                fr = fr_t.data_ptr()

            def to_dev(a, shape):
                if a is None:
                    return 0
                t = torch.as_tensor(a).to(device=dev, dtype=torch.float32).contiguous()
                if tuple(t.shape) != shape:
                    raise ValueError(f'expected shape {shape}, got {tuple(t.shape)}')
                keep.append(t)
                return t.data_ptr()
            nmix = self.n_classes if self.mode == 'RAW' else self.n_classes // 3
            if noise_mode == _NOISE_REFERENCE:
                if noise1 is not None or noise2 is not None:
                    raise ValueError("noise_mode='reference' draws its own noise: noise1 / noise2 must be None")
                noise1, noise2 = reference_noise(self.mode, rows, steps, self.n_classes, self.rnn_dims, self.aux_dims, dev)
                noise_mode = _cabi.NOISE_INJECTED
            n1 = to_dev(noise1, (steps, rows, nmix))
            n2 = to_dev(noise2, (steps, rows))
            xf = to_dev(x_forced, (steps, rows))
            xi = to_dev(x_init, (rows,))
            logits = torch.empty((steps, rows, self.n_classes), dtype=torch.float32, device=dev) if want_logits else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            want_kernel = self.kernel if kernel is None else kernel

            def launch(k):
                nat.generate(mels_t.data_ptr(), B, T, batched, target, overlap,
                             labels_ptr=labels.data_ptr(), samples_ptr=samples.data_ptr(), stream=stream,
                             noise_mode=noise_mode, seed=int(seed), noise1_ptr=n1, noise2_ptr=n2, x_forced_ptr=xf,
                             logits_ptr=logits.data_ptr() if logits is not None else 0,
                             kernel=k, x_init_ptr=xi, mels_padded=mels_padded,
                             frames_ptr=fr, batch_rows=batch_rows, team2_segment=team2_segment)
                return nat.last_timing()  # synchronises; surfaces device-side errors
            try:
                self.last_timing = launch(want_kernel)
            except _cabi.WrnnError as e:
                # WRNN_ERR_BUSY: a team kernel's 32 workgroups per XCD did not all become resident (another process holds CUs).  An explicit
                # kernel request fails as it is; AUTO retries once after a moment, then runs the any-shape kernel -- loudly.
                if e.code != _cabi.ERR_BUSY or want_kernel != _cabi.KERNEL_AUTO:
                    raise
                time.sleep(self.busy_retry_seconds)
                try:
                    self.last_timing = launch(_cabi.KERNEL_AUTO)
                except _cabi.WrnnError as e2:
                    if e2.code != _cabi.ERR_BUSY:
                        raise
                    self._warn_slow_path(f'the GPU is shared with another kernel ({e2})', steps)
                    self.last_timing = launch(_cabi.KERNEL_SIMPLE)
            if want_kernel == _cabi.KERNEL_AUTO and self.last_timing['kernel'] == _cabi.KERNEL_SIMPLE and not getattr(self, '_slow_warned', False):
                self._warn_slow_path(nat.team_info()[2] or 'the team kernels cannot run on this device', steps)
            frames_dev = keep[0] if ragged else None
            del keep
        return dict(samples=samples, labels=labels, logits=logits, rows=rows, steps=steps, frames=frames_dev)

    def _warn_slow_path(self, why: str, steps: int):
        """Once per model: AUTO is running ``WRNN_KERNEL_SIMPLE`` (one workgroup per row, weights streamed every step, ~1 ms per step:
        ~300x slower than the team kernels and slower than the reference on a few CPU cores)."""
        if getattr(self, '_slow_warned', False):
            return
        self._slow_warned = True
        warnings.warn(f'WaveRNN: generating on the any-shape fallback kernel (WRNN_KERNEL_SIMPLE), ~1 ms per sample step -- about 300x slower '
                      f'than the XCD-team kernels ({steps} steps: ~{steps * 1e-3:.0f} s per row).  Reason: {why}.', RuntimeWarning, stacklevel=3)

    def epilogue_device(self, res, batched, target, overlap, mu_law, wave_len):
        """float64 tail of generate() (:243-258) on the GPU (``wrnn_epilogue``): (wave_len,) float64 cuda tensor."""
        nat = self._native if self._native is not None else self.native()   # the tail reads no weights: no repack check
        dev = res['samples'].device
        with torch.cuda.device(dev):
            out = torch.empty((int(wave_len),), dtype=torch.float64, device=dev)
            nat.epilogue(res['samples'].data_ptr(), res['labels'].data_ptr(), res['rows'], res['steps'], batched, target,
                         overlap, mu_law, wave_len, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        return out

    def fold_target_for_device(self, n_frames, overlap, device=None, policy='per_xcd'):
        """``target`` for the extension values of ``generate``'s ``target`` argument (batched mode, one utterance):
        ``'per_xcd'``: as many folds (:293-340) as the device has 32-CU teams (8 on an MI355X), one per team on the latency kernel;
        ``'auto'``: the fold count with the lowest predicted loop time (``fold_plan``: 8 folds per team on the batch kernel for clips
        of a second or more -- 2.5-3x faster than ``'per_xcd'`` -- fewer for short clips, where 2 x overlap per fold dominates)."""
        dev = device if device is not None else next(self.parameters()).device
        n = _device_teams(dev)
        total = int(n_frames) * self.hop_length
        if policy == 'per_xcd':
            return fold_target(total, int(overlap), n)
        if policy != 'auto':
            raise ValueError(f"target must be an int, 'auto' or 'per_xcd', got {policy!r}")
        return fold_plan(total, int(overlap), n, self.mode, int(self.fold_min_target))[0]

    def generate(self, mels, save_path: Union[str, Path], batched, target, overlap, mu_law, epilogue='host',
                 **native_opts):
        """Same contract as the reference ``generate`` (:169-264), including its quirks:
        generates T*hop samples but returns (T-1)*hop (:184,:257); raises ``ValueError`` for T < 21
        (fade-out broadcast, :256-258); an unbatched call with B > 1 returns only utterance 0 (:253);
        batched mode needs B == 1 (:338); MOL forces ``mu_law=False`` (:174); the return value is float64
        (:245); the model is left in train mode (:262) and a wav is always written (:260).
        Sampling draws from a device counter RNG seeded from the global torch generator, so
        ``torch.manual_seed`` makes a call reproducible like it does for the reference; ``noise_mode='reference'`` instead replays
        the reference's OWN draws from that generator (``reference_noise``): ``torch.manual_seed(s)`` + this call returns what the
        reference's ``generate`` returns for seed s on a CPU model (labels bit-equal up to a near-tie of the sampler's race).
        ``target='auto'`` (extension, batched mode): the fold length with the lowest predicted latency for ONE utterance on this
        device (``fold_plan``; a 5 s clip: 64 folds of 2 265 steps on the batch kernel, ~18 ms); ``target='per_xcd'``: one fold per XCD
        team on the latency kernel (fewest crossfades that still use the whole chip, ~48 ms).  Crossfades as in the reference's batched mode.
        ``epilogue='device'`` runs decode / unfold / fade-out on the GPU (tables built like NumPy builds them;
        identical output up to the host libm's ``pow``) instead of the float64 NumPy pass on the host.
        """
        # (the reference calls self.eval() here, :170: BatchNorm on its running statistics.  The native prologue always folds the running
        # statistics, whatever the flag says, so the ~70-module walk is left out; the observable state -- train mode on return, :262 -- is kept)
        mu_law = mu_law if self.mode == 'RAW' else False
        start = time.time()
        mels_t = torch.as_tensor(mels)
        wave_len = (mels_t.size(-1) - 1) * self.hop_length
        if isinstance(target, str):
            target = self.fold_target_for_device(mels_t.size(-1), overlap, policy=target)
        if 'seed' not in native_opts and native_opts.get('noise_mode', _cabi.NOISE_PHILOX) in (_cabi.NOISE_PHILOX, 'philox'):
            native_opts['seed'] = int(torch.randint(0, 2 ** 62, (1,)).item())
        res = self.generate_raw(mels_t, batched, target, overlap, **native_opts)
        if self.verbose:
            self.gen_display(res['steps'] - 1, res['steps'], res['rows'], start)
        if epilogue == 'device':
            if wave_len < 20 * self.hop_length:   # the broadcast error of :258
                raise ValueError(f'operands could not be broadcast together with shapes ({max(wave_len, 0)},) '
                                 f'({20 * self.hop_length},) ({max(wave_len, 0)},)')
            output = self.epilogue_device(res, batched, target, overlap, mu_law, wave_len).cpu().numpy()
            save_wav(output, save_path, self.sample_rate)
            self.train()
            return output
        if epilogue != 'host':
            raise ValueError(f"epilogue must be 'host' or 'device', got {epilogue!r}")
        output = res['samples'].cpu().numpy().astype(np.float64)  # (rows, L)   :243-245

        if mu_law:
            output = decode_mu_law(output, self.n_classes, False)
        if batched:
            output = self.xfade_and_unfold(output, target, overlap)
        else:
            output = output[0]

        # Fade-out at the end to avoid signal cutting out suddenly   (:255-258)
        fade_out = np.linspace(1, 0, 20 * self.hop_length)
        output = output[:wave_len]
        output[-20 * self.hop_length:] *= fade_out

        save_wav(output, save_path, self.sample_rate)
        self.train()
        return output

    def generate_many(self, mels_list, save_paths=None, mu_law=True, epilogue='host', **native_opts):
        """Extension for serving loops: several independent utterances of different lengths in ONE device call, so that all
        8 XCD teams of the GPU work (a single unbatched utterance keeps one team = 1/8 of the chip busy; up to 8 utterances
        run on the latency kernel one per team, more on the batch kernel).  ``mels_list``: sequence of (n_mels, T_i) arrays.
        The clips are zero-padded on the right to the longest one -- exactly the padding ``generate`` itself applies
        (``pad_tensor``, :183) -- and handed over as a RAGGED batch (``frames_dev``): row i runs T_i * hop steps, the library
        orders the rows by length on the device and balances them over the XCD teams, so a short clip costs its own length, not
        the longest one's.  The first T_i * hop samples of row i are what a single ``generate`` call on clip i computes for the
        same noise.  Returns a list of float64 arrays, each what ``generate(mels_i[None], path_i, False, ...)`` returns
        ((T_i - 1) * hop samples, mu-law decoded, 20-hop fade-out); writes the wavs when ``save_paths`` is given.
        ``epilogue='device'``: decode / trim / fade-out of all rows in one ``wrnn_epilogue_rows`` launch."""
        self.eval()
        mu_law = mu_law if self.mode == 'RAW' else False
        arrs = [np.asarray(torch.as_tensor(m).detach().cpu().numpy(), dtype=np.float32) for m in mels_list]
        if not arrs or any(a.ndim != 2 or a.shape[0] != self.feat_dims for a in arrs):
            raise ValueError(f'expected a non-empty sequence of (n_mels={self.feat_dims}, T_i) arrays')
        lens = [a.shape[1] for a in arrs]
        if min(lens) < 21:   # the fade-out broadcast error of :258, raised before any device work
            t_bad = min(lens)
            raise ValueError(f'operands could not be broadcast together with shapes ({max((t_bad - 1) * self.hop_length, 0)},) '
                             f'({20 * self.hop_length},) ({max((t_bad - 1) * self.hop_length, 0)},)')
        tmax = max(lens)
        batch = np.zeros((len(arrs), self.feat_dims, tmax), np.float32)
        for i, a in enumerate(arrs):
            batch[i, :, :lens[i]] = a
        if 'seed' not in native_opts and native_opts.get('noise_mode', _cabi.NOISE_PHILOX) in (_cabi.NOISE_PHILOX, 'philox'):
            native_opts['seed'] = int(torch.randint(0, 2 ** 62, (1,)).item())
        ragged = len(set(lens)) > 1
        res = self.generate_raw(batch, False, 11000, 550, frames=np.asarray(lens, np.int32) if ragged else None, **native_opts)
        if epilogue == 'device':
            nat = self.native()
            dev = res['samples'].device
            wl_max = (tmax - 1) * self.hop_length
            with torch.cuda.device(dev):
                waves = torch.empty((len(arrs), wl_max), dtype=torch.float64, device=dev)
                nat.epilogue_rows(res['samples'].data_ptr(), res['labels'].data_ptr(), res['rows'], res['steps'], mu_law, wl_max,
                                  res['frames'].data_ptr() if ragged else 0, waves.data_ptr(), wl_max,
                                  torch.cuda.current_stream(dev).cuda_stream)
            waves = waves.cpu().numpy()
            outs = [waves[i, :(t_i - 1) * self.hop_length].copy() for i, t_i in enumerate(lens)]
        elif epilogue == 'host':
            samples = res['samples'].cpu().numpy().astype(np.float64)
            outs = []
            for i, t_i in enumerate(lens):
                wave_len = (t_i - 1) * self.hop_length
                out = samples[i, :t_i * self.hop_length]
                if mu_law:
                    out = decode_mu_law(out, self.n_classes, False)
                out = out[:wave_len]
                out[-20 * self.hop_length:] *= np.linspace(1, 0, 20 * self.hop_length)
                outs.append(out)
        else:
            raise ValueError(f"epilogue must be 'host' or 'device', got {epilogue!r}")
        if save_paths is not None:
            for out, path in zip(outs, save_paths):
                save_wav(out, path, self.sample_rate)
        self.train()
        return outs

    def gen_display(self, i, seq_len, b_size, start):
        """The reference's own ksamples/s meter (:267-271)."""
        gen_rate = (i + 1) / (time.time() - start) * b_size / 1000
        sys.stdout.write(f'\r| {(i + 1) * b_size}/{seq_len * b_size} | Batch Size: {b_size} | '
                         f'Gen Rate: {gen_rate:.1f}kHz | ')

    def xfade_and_unfold(self, y, target, overlap):
        """Equal-power crossfade + overlap-add of the folds (:342-405), float64 on the host."""
        num_folds, length = y.shape
        target = length - 2 * overlap
        total_len = num_folds * (target + overlap) + overlap
        silence_len = overlap // 2
        fade_len = overlap - silence_len
        ramp = np.linspace(-1, 1, fade_len, dtype=np.float64)
        fade_in = np.concatenate([np.zeros(silence_len, dtype=np.float64), np.sqrt(0.5 * (1 + ramp))])
        fade_out = np.concatenate([np.ones(silence_len, dtype=np.float64), np.sqrt(0.5 * (1 - ramp))])
        y[:, :overlap] *= fade_in
        y[:, -overlap:] *= fade_out
        unfolded = np.zeros(total_len, dtype=np.float64)
        for i in range(num_folds):
            lo = i * (target + overlap)
            unfolded[lo:lo + target + 2 * overlap] += y[i]
        return unfolded

    # ------------------------------------------------------------ bookkeeping
    def get_step(self):
        return self.step.data.item()

    def log(self, path, msg):
        with open(path, 'a') as f:
            print(msg, file=f)

    def load(self, path: Union[str, Path]):
        device = next(self.parameters()).device
        self.load_state_dict(torch.load(path, map_location=device), strict=False)

    def save(self, path: Union[str, Path]):
        torch.save(self.state_dict(), path)

    def num_params(self, print_out=True):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad) / 1_000_000
        if print_out:
            print('Trainable Parameters: %.3fM' % n)
        return n


class _LoopTrainFn(torch.autograd.Function):
    """loss = L(loop layers(x, mels_up, aux; params)) with forward and backward in ``wrnn_train_step``."""

    @staticmethod
    def forward(ctx, model, x, y, want_logits, mels_up, aux, *params):
        nat = model._native_handle()
        dev = x.device
        B, L = x.shape
        grads = [torch.empty_like(p) for p in params]
        d_m, d_a = torch.empty_like(mels_up), torch.empty_like(aux)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        logits = torch.empty((B, L, model.n_classes), dtype=torch.float32, device=dev) if want_logits else None
        ps = [p.detach().contiguous() for p in params]
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in grads], x.data_ptr(), mels_up.data_ptr(), aux.data_ptr(),
                           y.data_ptr(), B, L, loss.data_ptr(), logits.data_ptr() if want_logits else 0, d_m.data_ptr(), d_a.data_ptr(), st)
            model._train_generation = getattr(model, '_train_generation', 0) + 1   # the workspace now holds THIS pass (see _LoopForwardFn)
            if model.check_device_errors is True:
                nat.sync_status(st)   # waits for the stream: a busy GPU / a timed-out team kernel raises here, not as a silent NaN
        ctx.save_for_backward(d_m, d_a, *grads)
        if want_logits:
            ctx.mark_non_differentiable(logits)
            return loss, logits
        return (loss,)

    @staticmethod
    def backward(ctx, g_loss, *unused):
        d_m, d_a, *grads = ctx.saved_tensors
        return (None, None, None, None, d_m * g_loss, d_a * g_loss) + tuple(g * g_loss for g in grads)


class _LoopForwardFn(torch.autograd.Function):
    """y_hat = loop layers(x, mels_up, aux; params): ``wrnn_train_forward`` / ``wrnn_train_backward`` (activations stay in the handle's
    workspace between the two, so backward must be the next training call on the model -- what a training loop does)."""

    @staticmethod
    def forward(ctx, model, x, mels_up, aux, *params):
        nat = model._native_handle()
        dev = x.device
        B, L = x.shape
        logits = torch.empty((B, L, model.n_classes), dtype=torch.float32, device=dev)
        ps = [p.detach().contiguous() for p in params]
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            nat.train_forward([p.data_ptr() for p in ps], x.data_ptr(), mels_up.data_ptr(), aux.data_ptr(), B, L, logits.data_ptr(), st)
            if model.check_device_errors is True:
                nat.sync_status(st)
        # the activations of this pass live in the handle's ONE workspace: stamp the pass, so that a backward that comes after another
        # training call on the model (a second forward, a validation batch through training_loss, gradient accumulation over two forwards)
        # fails loudly instead of differentiating the other pass (round-3 advisor finding)
        model._train_generation = getattr(model, '_train_generation', 0) + 1
        ctx.generation = model._train_generation
        ctx.model = model
        ctx.save_for_backward(x, mels_up, aux, *ps)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        x, mels_up, aux, *ps = ctx.saved_tensors
        model = ctx.model
        if getattr(model, '_train_generation', 0) != ctx.generation:
            raise RuntimeError('WaveRNN: another training call on this model ran between this forward and its backward; the forward\'s '
                               'activations (kept in the native handle\'s single workspace) are gone.  Call backward() before the next '
                               'forward / training_loss, or use training_loss() (forward + backward in one native call).')
        nat = model._native_handle()
        dev = x.device
        B, L = x.shape
        grads = [torch.empty_like(p) for p in ps]
        d_m, d_a = torch.empty_like(mels_up), torch.empty_like(aux)
        dl = d_logits.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            nat.train_backward([p.data_ptr() for p in ps], [g.data_ptr() for g in grads], dl.data_ptr(), x.data_ptr(), mels_up.data_ptr(),
                               aux.data_ptr(), B, L, d_m.data_ptr(), d_a.data_ptr(), st)
            if model.check_device_errors is True:
                nat.sync_status(st)
        return (None, None, d_m, d_a) + tuple(grads)
