"""TEST INFRASTRUCTURE -- not product code.

Runs the *unmodified* reference ``WaveRNN.generate()`` (PyTorch, CPU) from
``/root/reference`` so that its outputs can be frozen as golden vectors
(``oracle/make_golden.py`` -> ``tests/golden/``) and used to validate the C
restatement in ``oracle/wavernn_oracle.c``.

``/root/reference`` exists only in the build container, never on the GPU box:
nothing under ``tests/ -m gpu``, ``bench.py`` or ``__graft_entry__.smoke()``
imports this file.

Non-semantic shims needed to import the reference under this image
(SURVEY.md section 8c):
  * ``librosa`` is not installed; on the generate path it is only touched by
    ``save_wav`` (``wavernn/utils/dsp.py:22-23``) -> stub that captures the
    float32 array instead of writing it.
  * ``np.cumproduct`` was removed in NumPy 2 (``fatchord_version.py:68``).
  * ``sys.dont_write_bytecode`` so the read-only tree is not touched;
    ``wavernn.utils.paths.Paths`` is never constructed (it ``os.makedirs``
    inside the reference tree, ``paths.py:19-21``).

Randomness: the reference draws through the global torch CPU generator.
``torch.multinomial(p, 1, True)`` on CPU is ``argmax(p / q)`` with
``q = empty_like(p).exponential_(1)`` (verified: every label of every golden fixture
is the reference's own ``multinomial`` draw, and ``tests/test_oracle_golden.py``
reproduces all of them from the replayed ``q``), so the noise the reference consumed for a
given ``torch.manual_seed`` is *replayed* here by re-seeding and issuing the
same ``exponential_`` / ``uniform_`` calls -- no monkeypatching of the
sampler.  ``torch.multinomial`` / the MOL sampler are wrapped only to *record*
what they return.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types
from typing import Dict, Optional

import numpy as np

REFERENCE_ROOT = os.environ.get('WRNN_REFERENCE_ROOT', '/root/reference')

_ref = None


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'wavernn', 'models', 'fatchord_version.py'))


def load_reference():
    """Import the reference package with the shims above; returns a namespace."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
    sys.dont_write_bytecode = True
    if not hasattr(np, 'cumproduct'):
        np.cumproduct = np.cumprod  # fatchord_version.py:68
    captured = {}
    if 'librosa' not in sys.modules:
        lib = types.ModuleType('librosa')

        def write_wav(path, x, sr=None):  # dsp.py:23
            captured['path'] = path
            captured['wav'] = np.array(x, copy=True)
            captured['sr'] = sr
        lib.output = types.SimpleNamespace(write_wav=write_wav)
        sys.modules['librosa'] = lib
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torch  # noqa: F401
    from wavernn.utils import hparams as hp
    if not hp.is_configured():
        hp.configure(os.path.join(REFERENCE_ROOT, 'wavernn_hparams.py'))
    import wavernn.models.fatchord_version as fv
    import wavernn.utils.distribution as dist
    import wavernn.utils.dsp as dsp
    _ref = types.SimpleNamespace(fv=fv, dist=dist, dsp=dsp, hp=hp, captured=captured)
    return _ref


def build_reference_model(state_dict: Dict[str, np.ndarray], mode: str = 'RAW', bits: int = 10,
                          **dims):
    """Instantiate the reference module and load a (numpy) state_dict strictly."""
    import torch
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    ref = load_reference()
    d = dict(DEFAULT_DIMS)
    d.update(dims)
    d['bits'] = bits
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.fv.WaveRNN(rnn_dims=d['rnn_dims'], fc_dims=d['fc_dims'], bits=d['bits'],
                               pad=d['pad'], upsample_factors=d['upsample_factors'],
                               feat_dims=d['feat_dims'], compute_dims=d['compute_dims'],
                               res_out_dims=d['res_out_dims'], res_blocks=d['res_blocks'],
                               hop_length=d['hop_length'], sample_rate=d['sample_rate'], mode=mode)
    sd = {k: torch.from_numpy(np.array(v, copy=True)) for k, v in state_dict.items()}
    model.load_state_dict(sd, strict=True)
    return model


def reference_generate(model, mels: np.ndarray, seed: int, batched: bool = False,
                       target: int = 11000, overlap: int = 550, mu_law: bool = True,
                       num_threads: Optional[int] = None) -> dict:
    """Run the unmodified ``generate()`` (fatchord_version.py:169-264).

    Returns dict(wav=float64 (wave_len,), saved=float32 array handed to
    save_wav, labels=int32 (L, B) [RAW] or samples=float32 (L, B) [MOL],
    seconds=wall time of generate()).
    """
    import time
    import torch
    ref = load_reference()
    rec = []
    if model.mode == 'RAW':
        orig = torch.multinomial

        def recording_multinomial(p, n, replacement=False, **kw):
            out = orig(p, n, replacement, **kw)
            rec.append(out.reshape(-1).to(torch.int32).clone())
            return out
        patch_target, patch_name, patched = torch, 'multinomial', recording_multinomial
    else:
        orig = ref.fv.sample_from_discretized_mix_logistic

        def recording_mol(y, log_scale_min=None):
            out = orig(y, log_scale_min)
            rec.append(out.reshape(-1).clone())
            return out
        patch_target, patch_name, patched = ref.fv, 'sample_from_discretized_mix_logistic', recording_mol

    old_threads = torch.get_num_threads()
    if num_threads is not None:
        torch.set_num_threads(num_threads)
    setattr(patch_target, patch_name, patched)
    try:
        torch.manual_seed(seed)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            wav = model.generate(torch.from_numpy(np.ascontiguousarray(mels)), '/dev/null/unused.wav',
                                 batched, target, overlap, mu_law)
        dt = time.perf_counter() - t0
    finally:
        setattr(patch_target, patch_name, orig)
        torch.set_num_threads(old_threads)
    out = dict(wav=np.asarray(wav, dtype=np.float64), saved=ref.captured.get('wav'), seconds=dt)
    stacked = torch.stack(rec).numpy()  # (L, B)
    if model.mode == 'RAW':
        out['labels'] = stacked.astype(np.int32)
    else:
        out['samples'] = stacked.astype(np.float32)
    return out


def reference_forward(model, x: np.ndarray, mels_padded: np.ndarray, y: np.ndarray) -> dict:
    """The unmodified ``forward`` (fatchord_version.py:131-167) in eval mode (BatchNorm on its running statistics, as
    in ``generate``) and the loss ``wavernn_train.py:82,112-121`` applies to it.  x (B, L) float32, mels_padded
    (B, n_mels, T + 2*pad), y (B, L) int64 labels (RAW) / float32 targets (MOL)."""
    import torch
    import torch.nn.functional as F
    ref = load_reference()
    model.eval()
    with torch.no_grad():
        y_hat = model(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(mels_padded)))

        def loss_of(yh, yy):
            yy = torch.from_numpy(np.ascontiguousarray(yy))
            if model.mode == 'RAW':
                return float(F.cross_entropy(yh.transpose(1, 2).unsqueeze(-1), yy.long().unsqueeze(-1)))
            return float(ref.dist.discretized_mix_logistic_loss(yh, yy.float().unsqueeze(-1)))
        out = dict(logits=y_hat.numpy(), loss=loss_of(y_hat, y), loss_of=loss_of)
    model.train()
    return out


def reference_train_step(model, x: np.ndarray, mels_padded: np.ndarray, y: np.ndarray) -> dict:
    """One iteration of the reference's training loop body (wavernn_train.py:103-122) on the unmodified module in train()
    mode: y_hat = model(x, m), the loss, loss.backward().  Returns the loss, the fc3 outputs and every parameter gradient."""
    import torch
    import torch.nn.functional as F
    ref = load_reference()
    model.train()
    model.zero_grad()
    xt = torch.from_numpy(np.ascontiguousarray(x))
    mt = torch.from_numpy(np.ascontiguousarray(mels_padded))
    yt = torch.from_numpy(np.ascontiguousarray(y))
    y_hat = model(xt, mt)
    if model.mode == 'RAW':
        loss = F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), yt.long().unsqueeze(-1))
    else:
        loss = ref.dist.discretized_mix_logistic_loss(y_hat, yt.float().unsqueeze(-1))
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
    return dict(loss=float(loss.detach()), logits=y_hat.detach().numpy(), grads=grads)


def replay_noise(seed: int, mode: str, steps: int, rows: int, n_classes: int = 1024,
                 rnn_dims: int = 512, aux_dims: int = 32) -> dict:
    """See oracle/noise.py (kept there so the GPU box can replay without the reference)."""
    from oracle.noise import noise_from_seed
    return noise_from_seed(seed, mode, steps, rows, n_classes, rnn_dims, aux_dims)


def reference_upsample(model, mels: np.ndarray):
    """pad_tensor + UpsampleNetwork exactly as generate() calls them (:183-186)."""
    import torch
    model.eval()
    with torch.no_grad():
        m = torch.from_numpy(np.ascontiguousarray(mels))
        m = model.pad_tensor(m.transpose(1, 2), pad=model.pad, side='both')
        up, aux = model.upsample(m.transpose(1, 2))
    model.train()
    return up.numpy(), aux.numpy()


def reference_dm_generate(state_dict: Dict[str, np.ndarray], seq_len: int, seed: int) -> dict:
    """Run the reference's ``deepmind_version.WaveRNN.generate(seq_len)`` (:75-165).  Upstream it cannot run as
    shipped: it calls ``stream(fmt, args)`` with two arguments (:159) while ``display.stream`` takes one -- that
    progress print is replaced by a no-op (non-semantic); everything else is the unmodified reference."""
    import torch
    load_reference()
    import wavernn.models.deepmind_version as dm
    dm.stream = lambda *a, **k: None
    with contextlib.redirect_stdout(io.StringIO()):
        model = dm.WaveRNN(hidden_size=state_dict['R.weight'].shape[1], quantisation=state_dict['O2.weight'].shape[0])
    model.load_state_dict({k: torch.from_numpy(np.array(v, copy=True)) for k, v in state_dict.items()}, strict=True)
    torch.manual_seed(seed)
    output, coarse, fine = model.generate(seq_len)
    return dict(output=np.asarray(output), coarse=np.asarray(coarse), fine=np.asarray(fine))
