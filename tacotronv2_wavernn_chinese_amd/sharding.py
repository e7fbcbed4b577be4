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


def balanced_shards(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced assignment ("length-balanced bins since loop time is proportional to T"): longest-first
    greedy bin packing (LPT) -- every clip, longest first, goes to the rank with the smallest total so far (ties: the lower
    rank).  Deterministic, so every rank computes the same table without talking to the others.  Returns one index list per
    rank, each in input order.  With equal lengths this is the round-robin assignment."""
    if world < 1:
        raise ValueError('world must be >= 1')
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    bins: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], len(bins[k]), k))
        bins[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(b) for b in bins]


def generate_sharded(generate_one: Optional[Callable[[int, np.ndarray], np.ndarray]], mels: Sequence[np.ndarray],
                     gather: bool = True, balance: bool = False,
                     generate_many: Optional[Callable[[List[int], List[np.ndarray]], Sequence[np.ndarray]]] = None
                     ) -> Optional[List[np.ndarray]]:
    """Run the vocoder on this rank's share of ``mels`` and gather all results on every rank.

    ``generate_one(index, mel)`` is typically ``lambda i, m: model.generate(m[None], path_i, False, target, overlap, mu_law)``;
    ``generate_many(indices, mels)`` (takes precedence) hands the rank's whole share to one call -- typically
    ``lambda idx, ms: model.generate_many(ms)``, one ragged device call that keeps all XCD teams of the GPU busy.
    ``balance``: assign clips by length (``balanced_shards`` over the frame counts ``mel.shape[-1]``) instead of round-robin.
    Returns the wavs in input order (None when ``gather`` is False and there is more than one rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    if balance:
        mine = balanced_shards([int(np.shape(m)[-1]) for m in mels], world)[rank]
    else:
        mine = shard_indices(len(mels), rank, world)
    if generate_many is not None:
        outs = generate_many(list(mine), [mels[i] for i in mine]) if mine else []
        local = [(i, np.asarray(w)) for i, w in zip(mine, outs)]
    else:
        if generate_one is None:
            raise ValueError('generate_sharded needs generate_one or generate_many')
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
