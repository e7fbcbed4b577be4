"""TEST INFRASTRUCTURE -- not product code.

numpy/ctypes front-end of the C restatement ``oracle/wavernn_oracle.c`` plus a
numpy restatement of the float64 epilogue.  Imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  The
product package never imports this module.

Parity pin: see the header of ``wavernn_oracle.c`` -- pinned against golden
vectors minted from the unmodified reference by ``oracle/make_golden.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MODE_RAW, MODE_MOL = 0, 1
NOISE_EXPO, NOISE_ARGMAX = 0, 2


class _Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('rnn_dims', 'fc_dims', 'feat_dims', 'aux_dims', 'compute_dims',
                                       'res_out_dims', 'res_blocks', 'pad', 'n_up')] + \
               [('up', C.c_int * 4), ('n_classes', C.c_int), ('mode', C.c_int)]


_FP = C.POINTER(C.c_float)


class _Weights(C.Structure):
    _fields_ = [(n, _FP) for n in ('conv_in_w', 'bn0_w', 'bn0_b', 'bn0_m', 'bn0_v', 'res_conv1_w',
                                   'res_conv2_w', 'res_bn1', 'res_bn2', 'conv_out_w', 'conv_out_b')] + \
               [('up_w', _FP * 4)] + \
               [(n, _FP) for n in ('I_w', 'I_b', 'r1_wih', 'r1_whh', 'r1_bih', 'r1_bhh', 'r2_wih', 'r2_whh',
                                   'r2_bih', 'r2_bhh', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'fc3_w', 'fc3_b')]


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc); called by __graft_entry__.build()."""
    if force or not all(os.path.exists(os.path.join(_HERE, f))
                        for f in ('liboracle_base.so', 'liboracle_avx2.so')):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))


def _has_avx2() -> bool:
    try:
        with open('/proc/cpuinfo') as f:
            txt = f.read()
        return ' avx2 ' in txt and ' fma ' in txt
    except OSError:
        return False


_libs: Dict[str, C.CDLL] = {}
_HW_THREADS: Optional[int] = None


def _lib(fast: bool = False) -> C.CDLL:
    name = 'liboracle_avx2.so' if (fast and _has_avx2()) else 'liboracle_base.so'
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.wo_resnet.restype = C.c_int
        lib.wo_upsample.restype = C.c_int
        lib.wo_stretch_aux.restype = C.c_int
        lib.wo_fold.restype = C.c_int
        lib.wo_loop.restype = C.c_int
        lib.wo_num_threads.restype = C.c_int
        _libs[name] = lib
    return _libs[name]


def _p(a: Optional[np.ndarray], ty=C.c_float):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ty))


class OracleModel:
    """Holds a reference-layout state_dict for the C oracle."""

    def __init__(self, state_dict: Dict[str, np.ndarray], mode: str = 'RAW', bits: int = 10,
                 upsample_factors=(5, 5, 11), pad: int = 2, fast: bool = False):
        sd = state_dict
        g = lambda k: np.ascontiguousarray(np.asarray(sd[k], dtype=np.float32))
        self.mode = mode
        self.lib = _lib(fast)
        self._keep = []
        d = _Dims()
        d.rnn_dims = sd['rnn1.weight_hh_l0'].shape[1]
        d.fc_dims = sd['fc1.weight'].shape[0]
        d.compute_dims, d.feat_dims, ksz = sd['upsample.resnet.conv_in.weight'].shape
        assert ksz == 2 * pad + 1
        d.res_out_dims = sd['upsample.resnet.conv_out.weight'].shape[0]
        d.aux_dims = d.res_out_dims // 4
        d.res_blocks = sum(1 for k in sd if k.endswith('.conv1.weight'))
        d.pad = pad
        d.n_up = len(upsample_factors)
        for i, s in enumerate(upsample_factors):
            d.up[i] = int(s)
        d.n_classes = sd['fc3.weight'].shape[0]
        d.mode = MODE_RAW if mode == 'RAW' else MODE_MOL
        self.dims = d
        self.hop = int(np.prod(upsample_factors))
        w = _Weights()

        def put(field, arr):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            self._keep.append(arr)
            setattr(w, field, _p(arr))
        put('conv_in_w', g('upsample.resnet.conv_in.weight'))
        for f, s in (('bn0_w', 'weight'), ('bn0_b', 'bias'), ('bn0_m', 'running_mean'), ('bn0_v', 'running_var')):
            put(f, g('upsample.resnet.batch_norm.' + s))
        nb = d.res_blocks
        put('res_conv1_w', np.stack([g(f'upsample.resnet.layers.{i}.conv1.weight')[:, :, 0] for i in range(nb)]))
        put('res_conv2_w', np.stack([g(f'upsample.resnet.layers.{i}.conv2.weight')[:, :, 0] for i in range(nb)]))
        for f, bnn in (('res_bn1', 'batch_norm1'), ('res_bn2', 'batch_norm2')):
            put(f, np.stack([np.stack([g(f'upsample.resnet.layers.{i}.{bnn}.{s}')
                                       for s in ('weight', 'bias', 'running_mean', 'running_var')])
                             for i in range(nb)]))
        put('conv_out_w', g('upsample.resnet.conv_out.weight')[:, :, 0])
        put('conv_out_b', g('upsample.resnet.conv_out.bias'))
        for i in range(d.n_up):
            arr = np.ascontiguousarray(g(f'upsample.up_layers.{2 * i + 1}.weight').reshape(-1))
            self._keep.append(arr)
            w.up_w[i] = _p(arr)
        for f, k in (('I_w', 'I.weight'), ('I_b', 'I.bias'),
                     ('r1_wih', 'rnn1.weight_ih_l0'), ('r1_whh', 'rnn1.weight_hh_l0'),
                     ('r1_bih', 'rnn1.bias_ih_l0'), ('r1_bhh', 'rnn1.bias_hh_l0'),
                     ('r2_wih', 'rnn2.weight_ih_l0'), ('r2_whh', 'rnn2.weight_hh_l0'),
                     ('r2_bih', 'rnn2.bias_ih_l0'), ('r2_bhh', 'rnn2.bias_hh_l0'),
                     ('fc1_w', 'fc1.weight'), ('fc1_b', 'fc1.bias'), ('fc2_w', 'fc2.weight'),
                     ('fc2_b', 'fc2.bias'), ('fc3_w', 'fc3.weight'), ('fc3_b', 'fc3.bias')):
            put(f, g(k))
        self.w = w

    # ---- prologue -------------------------------------------------------
    def resnet(self, mels: np.ndarray) -> np.ndarray:
        """(B,F,T) -> aux per frame (B,T,R)."""
        mels = np.ascontiguousarray(mels, dtype=np.float32)
        B, F, T = mels.shape
        out = np.empty((B, T, self.dims.res_out_dims), dtype=np.float32)
        rc = self.lib.wo_resnet(C.byref(self.dims), C.byref(self.w), _p(mels), B, T, _p(out))
        assert rc == 0
        return out

    def upsample_mels(self, mels: np.ndarray) -> np.ndarray:
        """(B,F,T) -> (B,T*hop,F)."""
        mels = np.ascontiguousarray(mels, dtype=np.float32)
        B, F, T = mels.shape
        out = np.empty((B, T * self.hop, F), dtype=np.float32)
        rc = self.lib.wo_upsample(C.byref(self.dims), C.byref(self.w), _p(mels), B, T, _p(out))
        assert rc == 0
        return out

    def conditioning(self, mels: np.ndarray):
        """generate() :183-186 -> (mels_up (B,L,F), aux_up (B,L,R))."""
        auxf = self.resnet(mels)
        B, T, R = auxf.shape
        aux = np.empty((B, T * self.hop, R), dtype=np.float32)
        self.lib.wo_stretch_aux(_p(auxf), B, T, R, self.hop, _p(aux))
        return self.upsample_mels(mels), aux

    def fold(self, x: np.ndarray, target: int, overlap: int) -> np.ndarray:
        """fold_with_overlap (:293-340); x (1, L, feat)."""
        assert x.shape[0] == 1
        x = np.ascontiguousarray(x, dtype=np.float32)
        L, feat = x.shape[1], x.shape[2]
        n = self.lib.wo_fold(_p(x), C.c_long(L), feat, target, overlap, None)
        out = np.empty((n, target + 2 * overlap, feat), dtype=np.float32)
        self.lib.wo_fold(_p(x), C.c_long(L), feat, target, overlap, _p(out))
        return out

    # ---- loop -----------------------------------------------------------
    def loop(self, cond_m: np.ndarray, cond_a: np.ndarray, noise_mode: int = NOISE_EXPO,
             noise1: Optional[np.ndarray] = None, noise2: Optional[np.ndarray] = None,
             x_forced: Optional[np.ndarray] = None, want_logits: bool = False,
             num_threads: Optional[int] = None) -> dict:
        cond_m = np.ascontiguousarray(cond_m, dtype=np.float32)
        cond_a = np.ascontiguousarray(cond_a, dtype=np.float32)
        B, L, F = cond_m.shape
        assert cond_a.shape[:2] == (B, L)
        NC = self.dims.n_classes
        if noise1 is not None:
            noise1 = np.ascontiguousarray(noise1, dtype=np.float32)
            exp_shape = (L, B, NC) if self.mode == 'RAW' else (L, B, NC // 3)
            assert noise1.shape == exp_shape, (noise1.shape, exp_shape)
        elif self.mode == 'RAW':
            assert noise_mode == NOISE_ARGMAX
        if noise2 is not None:
            noise2 = np.ascontiguousarray(noise2, dtype=np.float32)
            assert noise2.shape == (L, B)
        if self.mode == 'MOL':
            assert noise1 is not None and noise2 is not None
        if x_forced is not None:
            x_forced = np.ascontiguousarray(x_forced, dtype=np.float32)
            assert x_forced.shape == (L, B)
        labels = np.empty((L, B), dtype=np.int32)
        samples = np.empty((L, B), dtype=np.float32)
        logits = np.empty((L, B, NC), dtype=np.float32) if want_logits else None
        margin = np.empty((L, B), dtype=np.float32)
        runner = np.empty((L, B), dtype=np.int32)
        # The per-step parallel regions are tiny (one matvec each): more than ~16 OpenMP threads only add fork/join and
        # barrier time -- on a many-core host (the GPU box exposes all its logical CPUs to the default team) the loop ran
        # ~10x slower than with 16 threads.  Default: min(hardware threads, 16).
        global _HW_THREADS
        if _HW_THREADS is None:
            _HW_THREADS = max(1, min(int(self.lib.wo_num_threads()), os.cpu_count() or 1))
        if num_threads is None:
            num_threads = min(_HW_THREADS, 16)
        self.lib.wo_set_num_threads(int(num_threads))
        rc = self.lib.wo_loop(C.byref(self.dims), C.byref(self.w), _p(cond_m), _p(cond_a), B, C.c_long(L),
                              noise_mode, _p(noise1), _p(noise2), _p(x_forced), _p(labels, C.c_int32),
                              _p(samples), _p(logits), _p(margin), _p(runner, C.c_int32))
        assert rc == 0
        return dict(labels=labels, samples=samples, logits=logits, margin=margin, runner=runner)

    def num_threads(self) -> int:
        """Hardware threads OpenMP would use by default (before any cap applied by ``loop``)."""
        global _HW_THREADS
        if _HW_THREADS is None:
            _HW_THREADS = max(1, min(int(self.lib.wo_num_threads()), os.cpu_count() or 1))
        return _HW_THREADS


class _DmModel(C.Structure):
    _fields_ = [('hidden', C.c_int), ('quant', C.c_int)] + \
               [(n, _FP) for n in ('R', 'O1w', 'O1b', 'O2w', 'O2b', 'O3w', 'O3b', 'O4w', 'O4b', 'Ic', 'If', 'bu', 'br', 'be')]


class DeepmindOracle:
    """C restatement of ``wavernn/models/deepmind_version.py`` generate() (rows A12 of SURVEY.md section 8a)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], fast: bool = False):
        self.lib = _lib(fast)
        self.lib.wo_dm_generate.restype = C.c_int
        self._keep = []
        m = _DmModel()
        m.hidden = state_dict['R.weight'].shape[1]
        m.quant = state_dict['O2.weight'].shape[0]
        for f, k in (('R', 'R.weight'), ('O1w', 'O1.weight'), ('O1b', 'O1.bias'), ('O2w', 'O2.weight'), ('O2b', 'O2.bias'),
                     ('O3w', 'O3.weight'), ('O3b', 'O3.bias'), ('O4w', 'O4.weight'), ('O4b', 'O4.bias'),
                     ('Ic', 'I_coarse.weight'), ('If', 'I_fine.weight'), ('bu', 'bias_u'), ('br', 'bias_r'), ('be', 'bias_e')):
            arr = np.ascontiguousarray(state_dict[k], dtype=np.float32)
            self._keep.append(arr)
            setattr(m, f, _p(arr))
        self.m = m
        self.quant = m.quant

    def generate(self, seq_len: int, noise: Optional[np.ndarray] = None) -> dict:
        """noise: (seq_len, 2, Q) Exp(1) draws (coarse, fine) or None for greedy."""
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32)
            assert noise.shape == (seq_len, 2, self.quant)
        coarse = np.empty(seq_len, np.int32)
        fine = np.empty(seq_len, np.int32)
        margin = np.empty((seq_len, 2), np.float32)
        runner = np.empty((seq_len, 2), np.int32)
        self.lib.wo_set_num_threads(min(max(1, min(int(self.lib.wo_num_threads()), os.cpu_count() or 1)), 16))
        rc = self.lib.wo_dm_generate(C.byref(self.m), C.c_long(seq_len), _p(noise), _p(coarse, C.c_int32), _p(fine, C.c_int32),
                                     _p(margin), _p(runner, C.c_int32))
        assert rc == 0
        # combine_signal, wavernn/utils/dsp.py:33-34
        return dict(coarse=coarse, fine=fine, output=coarse * 256 + fine - 2 ** 15, margin=margin, runner=runner)


# -------------------------------------------------------------- epilogue (f64)

def decode_mu_law(y: np.ndarray, mu: int) -> np.ndarray:
    """wavernn/utils/dsp.py:98-103 with from_labels=False (as :247-248 calls it)."""
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def xfade_and_unfold(y: np.ndarray, target: int, overlap: int) -> np.ndarray:
    """fatchord_version.py:342-405 (equal-power crossfade, overlap-add)."""
    y = np.array(y, dtype=np.float64, copy=True)
    num_folds, length = y.shape
    target = length - 2 * overlap
    total_len = num_folds * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.concatenate([np.zeros(silence_len), np.sqrt(0.5 * (1 + t))])
    fade_out = np.concatenate([np.ones(silence_len), np.sqrt(0.5 * (1 - t))])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    out = np.zeros(total_len, dtype=np.float64)
    for i in range(num_folds):
        s = i * (target + overlap)
        out[s:s + target + 2 * overlap] += y[i]
    return out


def epilogue(samples_bl: np.ndarray, n_classes: int, mu_law: bool, batched: bool, target: int,
             overlap: int, wave_len: int, hop_length: int) -> np.ndarray:
    """fatchord_version.py:243-258.  samples_bl: (B, L) fp32 values fed back."""
    out = np.asarray(samples_bl).astype(np.float64)
    if mu_law:
        out = decode_mu_law(out, n_classes)
    if batched:
        out = xfade_and_unfold(out, target, overlap)
    else:
        out = out[0]
    fade = np.linspace(1, 0, 20 * hop_length)
    out = out[:wave_len]
    out[-20 * hop_length:] *= fade  # raises ValueError for T < 21, like the reference
    return out


# ------------------------------------------------ losses on forward()'s output (wavernn_train.py:82,112-121)

def cross_entropy(y_hat: np.ndarray, y: np.ndarray) -> float:
    """F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), y.unsqueeze(-1)): mean over (B, L) of
    logsumexp(y_hat[b, t]) - y_hat[b, t, y[b, t]]; float32 per element, float64 mean."""
    yh = np.asarray(y_hat, np.float32).reshape(-1, y_hat.shape[-1])
    yy = np.asarray(y).reshape(-1).astype(np.int64)
    m = yh.max(axis=1, keepdims=True)
    lse = (m[:, 0] + np.log(np.exp(yh - m).sum(axis=1, dtype=np.float32))).astype(np.float32)
    return float((lse - yh[np.arange(yh.shape[0]), yy]).astype(np.float64).mean())


def _softplus(x):
    x = np.asarray(x, np.float32)
    with np.errstate(over='ignore'):
        return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0)))).astype(np.float32)   # F.softplus(beta=1, threshold=20)


def discretized_mix_logistic_loss(y_hat: np.ndarray, y: np.ndarray, num_classes: int = 65536,
                                  log_scale_min: float = float(np.log(1e-14))) -> float:
    """wavernn/utils/distribution.py:16-84 with reduce=True.  y_hat (B, L, 3*nr_mix), y (B, L) in [-1, 1]."""
    yh = np.asarray(y_hat, np.float32)
    nr = yh.shape[-1] // 3
    logit_probs, means = yh[..., :nr], yh[..., nr:2 * nr]
    log_scales = np.maximum(yh[..., 2 * nr:3 * nr], np.float32(log_scale_min))       # :31
    yy = np.asarray(y, np.float32)[..., None]
    cy = yy - means                                                                    # :36
    inv = np.exp(-log_scales)
    hb = np.float32(1.0 / (num_classes - 1))
    plus_in, min_in = inv * (cy + hb), inv * (cy - hb)
    sig = lambda v: (1.0 / (1.0 + np.exp(-v))).astype(np.float32)
    cdf_delta = sig(plus_in) - sig(min_in)                                             # :39-54
    log_cdf_plus = plus_in - _softplus(plus_in)                                        # :45
    log_one_minus_cdf_min = -_softplus(min_in)                                         # :49
    mid_in = inv * cy
    log_pdf_mid = mid_in - log_scales - 2.0 * _softplus(mid_in)                        # :57
    c2 = (cdf_delta > 1e-5).astype(np.float32)                                         # :68-72
    inner_inner = c2 * np.log(np.maximum(cdf_delta, np.float32(1e-12))) + (1.0 - c2) * (log_pdf_mid - np.float32(np.log((num_classes - 1) / 2)))
    c1 = (yy > 0.999).astype(np.float32)
    inner = c1 * log_one_minus_cdf_min + (1.0 - c1) * inner_inner
    c0 = (yy < -0.999).astype(np.float32)
    log_probs = c0 * log_cdf_plus + (1.0 - c0) * inner
    lm = logit_probs.max(axis=-1, keepdims=True)
    log_probs = log_probs + (logit_probs - (lm + np.log(np.exp(logit_probs - lm).sum(axis=-1, keepdims=True, dtype=np.float32))))   # :77
    m = log_probs.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(log_probs - m).sum(axis=-1, dtype=np.float32))     # log_sum_exp :6-12
    return float(-(lse.astype(np.float64)).mean())
