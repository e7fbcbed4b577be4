"""TEST / MEASUREMENT INFRASTRUCTURE -- not product code.

The reference's generate() loop issued op for op on PyTorch-CPU, for timing on a box that has no /root/reference (the GPU
box): same modules (`nn.GRUCell` built from the GRU layers' tensors as `get_gru_cell` does, `nn.Linear`), same per-step op
sequence (`cat`, `I`, two cells with residuals, three FC layers with the aux splits, `softmax`, `Categorical(...).sample()`
= `torch.multinomial`, the label -> [-1, 1] map), same shapes (wavernn/models/fatchord_version.py:194-237, :273-279).  It is
what `wavernn_gen.py` spends its time in; the figure is reported by bench.py as `cpu_baseline` with kind
"reference-ops (torch CPU, this box)".  The unmodified reference itself is timed by oracle/time_reference.py where its tree
exists (profiles/cpu_reference_container.json); tests/test_oracle_golden.py::test_torch_cpu_loop_is_the_reference_loop pins
this restatement to the reference's own labels.

Only bench.py's cpu_baseline leg and tests/ import this file.
"""
from __future__ import annotations

import time
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import torch_ref


def _cell(sd, name: str, in_dim: int, hid: int) -> nn.GRUCell:
    cell = nn.GRUCell(in_dim, hid)                                   # get_gru_cell (:273-279): the GRU layer's own tensors
    cell.weight_ih.data = sd[f'{name}.weight_ih_l0']
    cell.weight_hh.data = sd[f'{name}.weight_hh_l0']
    cell.bias_ih.data = sd[f'{name}.bias_ih_l0']
    cell.bias_hh.data = sd[f'{name}.bias_hh_l0']
    return cell


def _linear(sd, name: str) -> nn.Linear:
    w = sd[f'{name}.weight']
    lin = nn.Linear(w.shape[1], w.shape[0])
    lin.weight.data = w
    lin.bias.data = sd[f'{name}.bias']
    return lin


def run(state_dict: Dict[str, np.ndarray], mels: np.ndarray, steps: int, threads: int, seed: int = 42, pad: int = 2,
        max_seconds: float = 1e9) -> dict:
    """RAW mode, unbatched: `steps` loop steps of the clip `mels` (B, 80, T) on `threads` CPU threads.  Returns the labels
    (steps, B) and the seconds the loop took (the upsampling prologue is timed separately).  `max_seconds` bounds the loop: it stops
    early (and says how many steps it made) -- a measurement leg must not run away on a box where the thread count suits it badly."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, int(threads)))
    try:
        sd = {k: torch.from_numpy(np.array(v, copy=True)) for k, v in state_dict.items()}
        rnn_dims = sd['rnn1.weight_hh_l0'].shape[1]
        n_classes = sd['fc3.weight'].shape[0]
        aux_dims = sd['rnn2.weight_ih_l0'].shape[1] - rnn_dims
        I, fc1, fc2, fc3 = _linear(sd, 'I'), _linear(sd, 'fc1'), _linear(sd, 'fc2'), _linear(sd, 'fc3')
        out_labels = []
        with torch.no_grad():
            # the reference builds its two GRUCells INSIDE generate() (:188-189), behind the caller's manual_seed: their default
            # initialisation draws from the global generator before the first sample does (oracle/noise.py replays the same)
            torch.manual_seed(seed)
            rnn1, rnn2 = _cell(sd, 'rnn1', rnn_dims, rnn_dims), _cell(sd, 'rnn2', rnn_dims + aux_dims, rnn_dims)
            t0 = time.perf_counter()
            m = torch.from_numpy(np.ascontiguousarray(mels))
            m = F.pad(m, (pad, pad))                                   # pad_tensor(side='both') (:183, :281-291)
            up, aux = torch_ref.upsample(sd, m, pad=pad, training=False)
            B, L, _ = up.shape
            h1 = torch.zeros(B, rnn_dims)
            h2 = torch.zeros(B, rnn_dims)
            x = torch.zeros(B, 1)
            # the four aux slices the layers are conditioned on (:198-199), per step below: a[k][:, i, :]
            a = [aux[:, :, aux_dims * k:aux_dims * (k + 1)] for k in range(4)]
            t1 = time.perf_counter()
            feed = x
            for i in range(min(int(steps), L)):
                if (i & 31) == 0 and time.perf_counter() - t1 > max_seconds:
                    break
                v = I(torch.cat((feed, up[:, i, :], a[0][:, i, :]), dim=1))          # :203-209
                h1 = rnn1(v, h1)                                                    # :210-211
                v = v + h1
                h2 = rnn2(torch.cat((v, a[1][:, i, :]), dim=1), h2)                 # :213-214
                v = v + h2
                v = F.relu(fc1(torch.cat((v, a[2][:, i, :]), dim=1)))               # :216-218
                v = F.relu(fc2(torch.cat((v, a[3][:, i, :]), dim=1)))               # :219-221
                probs = F.softmax(fc3(v), dim=1)                                    # :223, :231
                label = torch.distributions.Categorical(probs).sample()             # :233-234 (= torch.multinomial)
                out_labels.append(label)
                feed = (2 * label.float() / (n_classes - 1.) - 1.).unsqueeze(-1)    # :235-237
            t2 = time.perf_counter()
        return dict(labels=torch.stack(out_labels).numpy(), loop_seconds=t2 - t1, prologue_seconds=t1 - t0, steps=len(out_labels), threads=threads)
    finally:
        torch.set_num_threads(old)
