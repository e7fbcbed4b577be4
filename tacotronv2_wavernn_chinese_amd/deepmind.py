"""Drop-in host side of the *secondary* WaveRNN of the reference: the unconditioned dual-softmax (coarse/fine)
model of ``wavernn/models/deepmind_version.py`` (``WaveRNN(hidden_size=896, quantisation=256)`` :9-31,
``generate(seq_len)`` :75-165 -> ``(output, coarse, fine)``).  No reference script imports that model (and its
``generate`` crashes upstream on a two-argument ``stream(...)`` call, :159); it is carried for coverage of
SURVEY.md section 8a row A12.  Parameters live in an ``nn.Module`` with the reference's keys and default
initialisation; the per-sample loop runs in libwavernn_amd.so (``csrc/loop_deepmind.hip``)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _cabi


class WaveRNN(nn.Module):
    def __init__(self, hidden_size=896, quantisation=256):
        super().__init__()
        self.hidden_size = hidden_size
        self.split_size = hidden_size // 2
        self.quantisation = quantisation
        self.R = nn.Linear(self.hidden_size, 3 * self.hidden_size, bias=False)
        self.O1 = nn.Linear(self.split_size, self.split_size)
        self.O2 = nn.Linear(self.split_size, quantisation)
        self.O3 = nn.Linear(self.split_size, self.split_size)
        self.O4 = nn.Linear(self.split_size, quantisation)
        self.I_coarse = nn.Linear(2, 3 * self.split_size, bias=False)
        self.I_fine = nn.Linear(3, 3 * self.split_size, bias=False)
        self.bias_u = nn.Parameter(torch.zeros(self.hidden_size))
        self.bias_r = nn.Parameter(torch.zeros(self.hidden_size))
        self.bias_e = nn.Parameter(torch.zeros(self.hidden_size))
        self.num_params()
        self._native: Optional[_cabi.NativeDeepmind] = None
        self._native_key = None

    def forward(self, prev_y, prev_hidden, current_coarse):
        raise NotImplementedError('training forward (deepmind_version.py:37-72) is out of scope of the mel->wav path')

    def _native_handle(self) -> _cabi.NativeDeepmind:
        dev = next(self.parameters()).device
        if dev.type == 'cuda':
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
        elif torch.cuda.is_available():
            idx = torch.cuda.current_device()
        else:
            raise RuntimeError('deepmind WaveRNN.generate needs an MI355X (HIP) device: there is no CPU fallback')
        key = (idx,) + tuple((id(p), p._version) for p in self.parameters())
        if self._native is None or self._native.device != idx:
            if self._native is not None:
                self._native.close()
            self._native = _cabi.NativeDeepmind(self.hidden_size, self.quantisation, idx)
            self._native_key = None
        if self._native_key != key:
            self._native.load_weights({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})
            self._native_key = key
        return self._native

    def generate(self, seq_len, *, noise_mode=_cabi.NOISE_PHILOX, seed=None, noise=None, kernel=0):
        """Returns ``(output, coarse, fine)`` like the reference (:161-165): int64 arrays of length ``seq_len``,
        ``output = coarse * 256 + fine - 2**15`` (``combine_signal``, wavernn/utils/dsp.py:33-34).
        ``kernel``: 0 auto, 1 single-workgroup kernel (reference-ordered sums), 2 team kernel (32 CUs of one XCD)."""
        nat = self._native_handle()
        nat.set_kernel(kernel)
        dev = torch.device('cuda', nat.device)
        with torch.cuda.device(dev):
            coarse = torch.empty(seq_len, dtype=torch.int32, device=dev)
            fine = torch.empty(seq_len, dtype=torch.int32, device=dev)
            nptr, keep = 0, None
            if noise is not None:
                keep = torch.as_tensor(noise).to(device=dev, dtype=torch.float32).contiguous()
                if tuple(keep.shape) != (seq_len, 2, self.quantisation):
                    raise ValueError(f'expected noise shaped {(seq_len, 2, self.quantisation)}, got {tuple(keep.shape)}')
                nptr, noise_mode = keep.data_ptr(), _cabi.NOISE_INJECTED
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            nat.generate(seq_len, coarse.data_ptr(), fine.data_ptr(), torch.cuda.current_stream(dev).cuda_stream,
                         noise_mode=noise_mode, seed=seed, noise_ptr=nptr)
            torch.cuda.synchronize(dev)
        c = coarse.cpu().numpy().astype(np.int64)
        f = fine.cpu().numpy().astype(np.int64)
        return c * 256 + f - 2 ** 15, c, f

    def get_initial_hidden(self, batch_size=1):
        return torch.zeros(batch_size, self.hidden_size, device=next(self.parameters()).device)

    def num_params(self, print_out=True):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad) / 1_000_000
        if print_out:
            print('Trainable Parameters: %.3f million' % n)
        return n
