"""TEST INFRASTRUCTURE -- replay of the sampling noise the reference consumes.

``generate()`` (fatchord_version.py:169-264) draws from the global torch CPU
generator.  After ``torch.manual_seed(seed)`` it consumes, in order:
  1. the default ``reset_parameters`` draws of two ``nn.GRUCell`` objects
     (``get_gru_cell`` :178-179, :273-279) -- discarded;
  2. RAW: one ``empty(rows, n_classes).exponential_(1)`` per step inside
     ``torch.multinomial(p, 1, True)`` (== ``argmax(p / q)``), :233-235;
     MOL: per step ``uniform_(1e-5, 1-1e-5)`` on (1, rows, 10) then on
     (1, rows) (``wavernn/utils/distribution.py:106,118``).
Needs only torch (no /root/reference), so it runs on the GPU box too.
"""
from __future__ import annotations

import numpy as np


def noise_from_seed(seed: int, mode: str, steps: int, rows: int, n_classes: int = 1024,
                    rnn_dims: int = 512, aux_dims: int = 32) -> dict:
    import torch
    torch.manual_seed(seed)
    torch.nn.GRUCell(rnn_dims, rnn_dims)             # get_gru_cell(self.rnn1)
    torch.nn.GRUCell(rnn_dims + aux_dims, rnn_dims)  # get_gru_cell(self.rnn2)
    if mode == 'RAW':
        expo = torch.empty(steps, rows, n_classes)
        for t in range(steps):
            expo[t] = torch.empty(rows, n_classes).exponential_(1)
        return dict(expo=expo.numpy())
    u_mix = torch.empty(steps, rows, 10)
    u_log = torch.empty(steps, rows)
    for t in range(steps):
        u_mix[t] = torch.empty(1, rows, 10).uniform_(1e-5, 1.0 - 1e-5)[0]
        u_log[t] = torch.empty(1, rows).uniform_(1e-5, 1.0 - 1e-5)[0]
    return dict(u_mix=u_mix.numpy(), u_log=u_log.numpy())


def noise_checksum(noise: dict) -> np.ndarray:
    parts = []
    for k in sorted(noise):
        a = noise[k]
        parts += [a.reshape(-1)[:8].astype(np.float64), [a.astype(np.float64).sum()]]
    return np.concatenate([np.asarray(p, dtype=np.float64).reshape(-1) for p in parts])


def dm_noise_from_seed(seed: int, steps: int, quant: int = 256) -> np.ndarray:
    """deepmind_version.generate() (:75-165) draws two Categorical samples per step (coarse :131, fine :151), each an
    ``empty(1, quant).exponential_(1)`` inside torch.multinomial; nothing else touches the generator."""
    import torch
    torch.manual_seed(seed)
    out = torch.empty(steps, 2, quant)
    for t in range(steps):
        out[t, 0] = torch.empty(1, quant).exponential_(1)[0]
        out[t, 1] = torch.empty(1, quant).exponential_(1)[0]
    return out.numpy()
