"""The losses the reference's training script applies to ``WaveRNN.forward``'s output (``wavernn_train.py:82,112-121``),
evaluated on the MI355X by ``wrnn_loss`` (csrc/losses.hip): ``F.cross_entropy`` for RAW models and
``discretized_mix_logistic_loss`` (``wavernn/utils/distribution.py:16-84``) for MOL models.  Forward values only."""
from __future__ import annotations

import torch


def voc_loss(model, y_hat: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``y_hat`` (B, L, n_classes): what ``model.forward(x, mels)`` returns; ``y`` (B, L): int class labels (RAW) or float
    targets in [-1, 1] (MOL) -- the tensors of ``wavernn_train.py:103-118`` before its ``unsqueeze`` / ``transpose``
    reshaping, which only exists to fit torch's loss signatures.  Returns a 0-dim float32 tensor on the model's device."""
    nat = model.native()
    dev = torch.device('cuda', nat.device)
    with torch.cuda.device(dev):
        yh = torch.as_tensor(y_hat).to(device=dev, dtype=torch.float32).contiguous()
        if yh.dim() != 3 or yh.size(-1) != model.n_classes:
            raise ValueError(f'expected y_hat (B, L, {model.n_classes}), got {tuple(yh.shape)}')
        yt = torch.as_tensor(y).to(device=dev)
        if tuple(yt.shape) != tuple(yh.shape[:2]):
            raise ValueError(f'expected y {tuple(yh.shape[:2])}, got {tuple(yt.shape)}')
        yt = (yt.to(torch.int32) if model.mode == 'RAW' else yt.to(torch.float32)).contiguous()
        out = torch.empty((), dtype=torch.float32, device=dev)
        nat.loss(yh.data_ptr(), yt.data_ptr(), yh.shape[0] * yh.shape[1], out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    return out


def discretized_mix_logistic_loss(model, y_hat, y):
    """Name-compatible entry for MOL models (``distribution.py:16``); ``y`` may carry the trailing unit axis the
    reference's loop adds (``wavernn_train.py:118``)."""
    y = torch.as_tensor(y)
    return voc_loss(model, y_hat, y.squeeze(-1) if y.dim() == 3 else y)
