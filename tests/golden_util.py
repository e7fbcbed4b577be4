"""Shared fixture loading for the parity tests (CPU oracle and GPU path)."""
from __future__ import annotations

import functools
import os

import numpy as np
import pytest

from oracle.noise import noise_checksum, noise_from_seed
from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# BASELINE-size fixtures (labels + wav only, oracle/make_golden.py LONG_CASES): their own tests, not the per-case sweeps
LONG_CASES = ['raw_peaky_b1_t401', 'raw_peaky_b8_t60']
ALL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith('.npz') and f[:4] in ('raw_', 'mol_')
                   and f[:-4] not in LONG_CASES)   # dm_*: tests/test_deepmind.py
RAW_CASES = [c for c in ALL_CASES if c.startswith('raw_')]
MOL_CASES = [c for c in ALL_CASES if c.startswith('mol_')]


@functools.lru_cache(maxsize=None)
def load_case(name: str) -> dict:
    """Fixture + rebuilt inputs (weights, mels, noise) for one golden case."""
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    fx = {k: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files}
    fx['name'] = name
    fx['state_dict'] = make_state_dict(int(fx['weight_seed']), mode=fx['mode'], variant=fx['variant'],
                                       bits=int(fx['bits']))
    fx['mels'] = make_mels(int(fx['mel_seed']), int(fx['B']), int(fx['T']))
    key = 'labels' if fx['mode'] == 'RAW' else 'samples'
    L, rows = fx[key].shape
    noise = noise_from_seed(int(fx['noise_seed']), fx['mode'], L, rows)
    if not np.allclose(noise_checksum(noise), fx['noise_checksum'], rtol=0, atol=0):
        pytest.skip('torch CPU RNG stream differs from the one the goldens were minted with')
    fx['noise'] = noise
    fx['L'], fx['rows'] = L, rows
    return fx
