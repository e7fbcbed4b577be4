"""Utterance-parallel generation over the GPUs of one node.

The mel->wav path has no exchange step: every clip starts from zero state
(``fatchord_version.py:194-196``), so clips shard across ranks with **no data-path collective**
(SURVEY.md section 8e).  One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over
xGMI on the GPU box, ``gloo`` in CPU tests): rank r generates clips r, r + N, ... with its own
``wrnn_handle``; the only communication is the final gather of the (ragged) wavs, O(100 KB) per clip.

The reference has nothing equivalent (its only parallel construct is the dead training-time
``data_parallel_workaround``, ``wavernn/utils/__init__.py:22-36``).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment: clip i -> rank i mod world (clips of one batch have similar length)."""
    return list(range(rank, n_items, world))


def generate_sharded(generate_one: Callable[[int, np.ndarray], np.ndarray], mels: Sequence[np.ndarray],
                     gather: bool = True) -> Optional[List[np.ndarray]]:
    """Run ``generate_one(index, mel)`` for this rank's share of ``mels`` and gather all results on every rank.

    ``generate_one`` is typically ``lambda i, m: model.generate(m[None], path_i, False, target, overlap, mu_law)``.
    Returns the wavs in input order (None on ranks != 0 when ``gather`` is False)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    mine = shard_indices(len(mels), rank, world)
    local = [(i, np.asarray(generate_one(i, mels[i]))) for i in mine]
    if world == 1:
        out: List[Optional[np.ndarray]] = [None] * len(mels)
        for i, w in local:
            out[i] = w
        return out  # type: ignore[return-value]
    if not gather:
        dist.barrier()
        return None
    # ragged wavs: one object gather (lengths differ per clip; total volume is tiny vs one xGMI link-second)
    bucket: List[Optional[list]] = [None] * world
    dist.all_gather_object(bucket, local)
    out = [None] * len(mels)
    for part in bucket:
        for i, w in part:  # type: ignore[union-attr]
            out[i] = w
    return out  # type: ignore[return-value]
