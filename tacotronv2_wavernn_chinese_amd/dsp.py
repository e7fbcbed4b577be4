"""Host-side DSP helpers on the generate() path.

Mirror of the subset of ``wavernn/utils/dsp.py`` the mel->wav path touches:
``label_2_float`` (:8-9), ``save_wav`` (:22-23), ``decode_mu_law`` (:98-103).
Feature extraction (mel/STFT/Griffin-Lim) is out of scope.
"""
from __future__ import annotations

import math

import numpy as np


def label_2_float(x, bits):
    return 2 * x / (2 ** bits - 1.) - 1.


def decode_mu_law(y, mu, from_labels=True):
    if from_labels:
        y = label_2_float(y, math.log2(mu))
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def save_wav(x, path, sample_rate=None):
    """float32 WAV at hp.sample_rate, what ``librosa.output.write_wav`` (librosa <= 0.7) produced."""
    from scipy.io import wavfile
    if sample_rate is None:
        from .hparams import hparams as hp
        sample_rate = hp.sample_rate
    wavfile.write(str(path), int(sample_rate), np.asarray(x).astype(np.float32))
