"""The caller of the training step: the vocoder training loop and its checkpoint files.

What ``wavernn_train.py:88-151`` does with a ``WaveRNN`` and a loader of ``(x, y, mels)`` batches (``collate_vocoder``,
``wavernn/utils/dataset.py:107-133``), and what ``wavernn/utils/checkpoints.py:28-78,81-138`` write next to it -- the same two
files per checkpoint (``*_weights.pyt`` = ``model.state_dict()``, ``*_optim.pyt`` = ``optimizer.state_dict()``) at the same
places (``logs_wavernn/checkpoints/latest_*.pyt`` + ``wave_step{k}K_*.pyt``, ``wavernn/utils/paths.py:11-17``), so a run can be
resumed by either code base and ``wavernn_gen.py`` finds the weights where it looks for them.

Out of scope here (data formats either side of the path only): the dataset reader.  ``train_set`` is any sized
iterable of ``(x, y, mels)``; ``collate_windows`` below builds such batches from in-memory ``(mel, quantised wav)`` pairs the
way the reference's collate cuts its windows, and is what the tests and ``--synthetic`` use.

The iteration itself is ``WaveRNN.training_loss`` (one ``wrnn_train_step``: forward, loss and backward of the loop layers on
the MI355X) unless a ``loss_func`` is given, in which case the loop body is the reference's own
``loss_func(model(x, m), y)`` through the differentiable ``forward()``.
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Callable, Iterable, Optional, Sequence, Tuple, Union

import numpy as np
import torch


class VocPaths:
    """The vocoder half of ``Paths`` (``wavernn/utils/paths.py:5-32``), rooted at ``base`` instead of the package's parent."""

    def __init__(self, base: Union[str, Path] = '.', create: bool = True):
        self.base = Path(base).expanduser().resolve()
        self.voc_checkpoints = self.base / 'logs_wavernn' / 'checkpoints'
        self.voc_latest_weights = self.voc_checkpoints / 'latest_weights.pyt'
        self.voc_latest_optim = self.voc_checkpoints / 'latest_optim.pyt'
        self.voc_output = self.base / 'logs_wavernn' / 'model_outputs'
        self.voc_log = self.voc_checkpoints / 'log.txt'
        if create:
            self.voc_checkpoints.mkdir(parents=True, exist_ok=True)
            self.voc_output.mkdir(parents=True, exist_ok=True)

    def named(self, name: Optional[str]) -> Tuple[Path, Path]:
        """(weights file, optimizer file) of the latest (``name=None``) or a named checkpoint."""
        if name is None:
            return self.voc_latest_weights, self.voc_latest_optim
        return self.voc_checkpoints / f'{name}_weights.pyt', self.voc_checkpoints / f'{name}_optim.pyt'


def _write_pair(files: Tuple[Path, Path], model, optimizer):
    have = [f.exists() for f in files]
    if have[0] != have[1]:          # checkpoints.py:45-49: a half-written checkpoint is an error, not something to overwrite
        raise FileNotFoundError(f'checkpoint {files[0].name} / {files[1].name}: exactly one of the two files exists')
    files[0].parent.mkdir(parents=True, exist_ok=True)
    model.save(files[0])
    torch.save(optimizer.state_dict(), files[1])


def save_checkpoint(paths: VocPaths, model, optimizer, *, name: Optional[str] = None):
    """``save_checkpoint('voc', ...)`` (checkpoints.py:28-78): always refresh ``latest_*``; with ``name`` also write the named pair."""
    _write_pair(paths.named(None), model, optimizer)
    if name:
        _write_pair(paths.named(name), model, optimizer)


def restore_checkpoint(paths: VocPaths, model, optimizer, *, name: Optional[str] = None, create_if_missing: bool = False):
    """``restore_checkpoint('voc', ...)`` (checkpoints.py:81-138).  Call after ``model.to(device)``: the optimizer state lands
    on the device of the parameters it belongs to."""
    w, o = paths.named(name)
    if w.exists() and o.exists():
        model.load(w)
        optimizer.load_state_dict(torch.load(o, map_location=next(model.parameters()).device))
    elif create_if_missing:
        save_checkpoint(paths, model, optimizer, name=name)
    else:
        raise FileNotFoundError(f'no {"named" if name else "latest"} checkpoint at {w}')


def label_2_float(x, bits: int):
    """dsp.py:36 of the reference: class index -> [-1, 1]."""
    return 2.0 * x / (2 ** bits - 1.0) - 1.0


def collate_windows(batch: Sequence[Tuple[np.ndarray, np.ndarray]], *, mode: str, bits: int, hop_length: int, pad: int,
                    seq_len: int, rng: np.random.Generator):
    """One training batch from ``(mel (n_mels, frames), quantised wav (frames * hop,) of class indices)`` pairs: a random window of
    ``seq_len`` samples + the ``pad`` context frames either side per pair (dataset.py:107-133).  Returns (x, y, mels) as the loop
    wants them: x (B, seq_len) floats in [-1, 1], y (B, seq_len) the next-sample targets (int64 classes for RAW, floats of a
    16-bit signal for MOL), mels (B, n_mels, seq_len / hop + 2 * pad)."""
    if seq_len % hop_length:
        raise ValueError('seq_len must be a multiple of hop_length')
    win = seq_len // hop_length + 2 * pad
    mels, labels = [], []
    for mel, wav in batch:
        room = mel.shape[-1] - 2 - (win + 2 * pad)
        if room <= 0:
            raise ValueError(f'an utterance of {mel.shape[-1]} frames is too short for a window of {win} (+{2 * pad + 2})')
        off = int(rng.integers(0, room))
        s0 = (off + pad) * hop_length
        mels.append(mel[:, off:off + win])
        labels.append(wav[s0:s0 + seq_len + 1])
    mels = torch.from_numpy(np.stack(mels).astype(np.float32))
    labels = torch.from_numpy(np.stack(labels).astype(np.int64))
    sig_bits = 16 if mode == 'MOL' else bits
    x = label_2_float(labels[:, :seq_len].float(), sig_bits)
    y = labels[:, 1:]
    if mode == 'MOL':
        y = label_2_float(y.float(), sig_bits)
    return x, y, mels


def voc_train_loop(paths: VocPaths, model, loss_func: Optional[Callable], optimizer, train_set: Iterable, test_set, lr: float,
                   total_steps: int, *, clip_grad_norm: Optional[float] = 4, checkpoint_every: int = 1000,
                   at_checkpoint: Optional[Callable] = None, report: Optional[Callable[[str], None]] = None):
    """``voc_train_loop`` (wavernn_train.py:88-151) on the MI355X.

    ``loss_func=None``: each iteration is ``model.training_loss(x, m, y)`` (forward + loss + backward of the loop layers fused
    in ``wrnn_train_step``).  With a ``loss_func`` (``F.cross_entropy`` / a MOL loss on torch tensors) the body is the
    reference's (:103-121): ``y_hat = model(x, m)``, the transpose / unsqueeze that fit torch's loss signatures, ``loss_func``.
    ``clip_grad_norm`` / ``checkpoint_every`` are ``hp.voc_clip_grad_norm`` / ``hp.voc_checkpoint_every``; ``at_checkpoint(model,
    test_set, step)`` stands where the reference calls ``gen_testset`` (:137-138); ``report`` receives the progress line
    (``stream(msg)``, :145).  Returns the list of per-iteration losses of this call.
    Deferred device errors (``check_device_errors='deferred'``): ``training_status()`` raises AFTER ``optimizer.step()`` of the failing
    iteration has run on whatever the step produced -- nothing corrupt reaches disk (no checkpoint is written behind a raised error), but the
    in-memory parameters and Adam moments are then poisoned: restore the latest checkpoint (``restore_checkpoint``) before continuing, do
    not just retry the iteration.
    """
    device = next(model.parameters()).device
    for g in optimizer.param_groups:
        g['lr'] = lr
    per_epoch = len(train_set)
    if per_epoch < 1:
        raise ValueError('voc_train_loop: the training set is empty')
    epochs = (total_steps - model.get_step()) // per_epoch + 1
    params = [p for p in model.parameters() if p.requires_grad]
    losses = []
    model.train()
    # Device-side errors of the training kernels are asked for once per iteration (at the loss.item() below, where the host waits anyway)
    # instead of inside every call, and the NaN message of the clipping waits for the same point: the host queues the upsample network's
    # backward, the clipping and Adam while the step is still running.  An error raises before the iteration's weights can reach a
    # checkpoint.
    check_mode, model.check_device_errors = getattr(model, 'check_device_errors', True), 'deferred'
    try:
        for epoch in range(1, epochs + 1):
            t0 = time.time()
            running = 0.0
            msg = ''
            for i, (x, y, m) in enumerate(train_set, 1):
                x, m, y = x.to(device), m.to(device), y.to(device)
                if loss_func is None:
                    loss = model.training_loss(x, m, y)
                else:
                    y_hat = model(x, m)
                    if model.mode == 'RAW':
                        y_hat = y_hat.transpose(1, 2).unsqueeze(-1)
                    else:
                        y = y.float()
                    loss = loss_func(y_hat, y.unsqueeze(-1))
                optimizer.zero_grad()
                loss.backward()
                norm = torch.nn.utils.clip_grad_norm_(params, clip_grad_norm) if clip_grad_norm is not None else None
                optimizer.step()
                value = loss.item()                  # the one host sync of an iteration (the reference has it too, :129)
                if hasattr(model, 'training_status'):
                    model.training_status()
                if norm is not None and not bool(torch.isfinite(norm)):
                    print('grad_norm was NaN!')
                losses.append(value)
                running += value
                step = model.get_step()
                if step % checkpoint_every == 0:
                    if at_checkpoint is not None:
                        at_checkpoint(model, test_set, step)
                        model.train()
                    save_checkpoint(paths, model, optimizer, name=f'wave_step{step // 1000}K')
                msg = (f'| Epoch: {epoch}/{epochs} ({i}/{per_epoch}) | Loss: {running / i:.4f} | '
                       f'{i / (time.time() - t0):.1f} steps/s | Step: {step // 1000}k | ')
                if report is not None:
                    report(msg)
            save_checkpoint(paths, model, optimizer)     # the optimizer state of the epoch's end, so resuming does not jump (:147-149)
            model.log(paths.voc_log, msg)
    finally:
        model.check_device_errors = check_mode
    return losses


# --------------------------------------------------------------------------- the data side of the loop
def read_feature_list(feature_path: Union[str, Path], *, seq_len: int, hop_length: int, pad: int, test_samples: int):
    """The training list ``get_vocoder_datasets`` reads (dataset.py:62-88): one ``wav.npy|...|mel.npy`` line per utterance (field 0
    the quantised wav, field 2 the (frames, n_mels) mel), utterances too short for one window dropped, ids shuffled with seed
    1234 and the last ``test_samples`` set aside.  Returns (train pairs, test pairs) of (wav path, mel path)."""
    import random
    win = seq_len // hop_length + 2 * pad
    items = []
    with open(feature_path, 'r', encoding='utf-8') as f:
        for line in f:
            parts = line.strip().split('|')
            if len(parts) < 3:
                continue
            wav, mel = parts[0].strip(), parts[2].strip()
            if np.load(mel, mmap_mode='r').shape[0] - (win + 2 * pad + 2) < 0:
                continue
            items.append((wav, mel))
    ids = list(range(len(items)))
    random.Random(1234).shuffle(ids)
    if test_samples:
        return [items[i] for i in ids[:-test_samples]], [items[i] for i in ids[-test_samples:]]
    return [items[i] for i in ids], []


class WindowLoader:
    """The train ``DataLoader`` of dataset.py:90-95 as the loop sees it: a sized iterable, shuffled per epoch, each batch a fresh
    random window per utterance (``collate_windows``).  ``pairs`` are (wav path, mel path) of ``.npy`` files or in-memory
    (mel (n_mels, frames), wav) arrays; a last short batch is kept, like the reference's loader."""

    def __init__(self, pairs, batch_size: int, *, mode: str, bits: int, hop_length: int, pad: int, seq_len: int, seed: int = 0):
        self.pairs, self.batch_size = list(pairs), int(batch_size)
        self.kw = dict(mode=mode, bits=bits, hop_length=hop_length, pad=pad, seq_len=seq_len)
        self.rng = np.random.Generator(np.random.PCG64(seed))

    def __len__(self):
        return (len(self.pairs) + self.batch_size - 1) // self.batch_size

    @staticmethod
    def _arrays(pair):
        a, b = pair
        if isinstance(a, (str, Path)):
            return np.load(b).T, np.load(a)           # VocoderDataset.__getitem__ (dataset.py:52-57): files hold (frames, n_mels)
        return a, b

    def __iter__(self):
        order = self.rng.permutation(len(self.pairs))
        for i in range(0, len(order), self.batch_size):
            yield collate_windows([self._arrays(self.pairs[j]) for j in order[i:i + self.batch_size]], rng=self.rng, **self.kw)


def synthetic_pairs(n: int, frames: int, *, bits: int, n_mels: int, hop_length: int, seed: int = 0):
    """``n`` in-memory (mel, quantised wav) pairs of a slow random walk: stands in for a corpus (there is none in this repository)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n):
        walk = np.cumsum(rng.integers(-3, 4, size=frames * hop_length)) + 2 ** (bits - 1)
        out.append((rng.random((n_mels, frames), dtype=np.float32), (walk % 2 ** bits).astype(np.int64)))
    return out


def main(argv=None):
    """``python wavernn_train.py`` (wavernn_train.py:20-86) on the MI355X."""
    import argparse
    import torch.nn.functional as F
    from .gen import build_model_from_hparams
    from .hparams import DEFAULT_HPARAMS, hparams as hp
    parser = argparse.ArgumentParser(description='Train WaveRNN Vocoder')
    parser.add_argument('--gta', '-g', action='store_true', help='accepted for compatibility: the list file names the mels to train on')
    parser.add_argument('--hp_file', metavar='FILE', default=DEFAULT_HPARAMS, help='The file to use for the hyperparameters')
    parser.add_argument('--synthetic', type=int, metavar='N', default=0, help='train on N synthetic utterances instead of hp.feature_path')
    parser.add_argument('--total_steps', type=int, help='override hp.voc_total_steps')
    parser.add_argument('--reference_body', action='store_true',
                        help="run the reference's loop body (model(x, m) + torch loss) instead of the fused training_loss")
    args = parser.parse_args(argv)
    hp.configure(args.hp_file)
    if not torch.cuda.is_available():
        raise RuntimeError('this vocoder trains on an MI355X only (no CPU path)')
    if int(np.prod(hp.voc_upsample_factors)) != hp.hop_length:
        raise ValueError('voc_upsample_factors must factorise hop_length')
    paths = VocPaths('.')
    model = build_model_from_hparams().to(torch.device('cuda'))
    optimizer = torch.optim.Adam(model.parameters())
    restore_checkpoint(paths, model, optimizer, create_if_missing=True)
    kw = dict(mode=hp.voc_mode, bits=hp.bits, hop_length=hp.hop_length, pad=hp.voc_pad, seq_len=hp.voc_seq_len)
    if args.synthetic:
        sig_bits = 16 if hp.voc_mode == 'MOL' else hp.bits
        train, test = synthetic_pairs(args.synthetic, 40, bits=sig_bits, n_mels=hp.num_mels, hop_length=hp.hop_length), []
    else:
        train, test = read_feature_list(hp.feature_path, seq_len=hp.voc_seq_len, hop_length=hp.hop_length, pad=hp.voc_pad,
                                        test_samples=hp.voc_test_samples)
    total = args.total_steps if args.total_steps is not None else hp.voc_total_steps
    print(f'Remaining {total - model.get_step()} steps | batch {hp.voc_batch_size} | lr {hp.voc_lr} | seq_len {hp.voc_seq_len} | '
          f'{len(train)} training utterances')
    loss_func = None
    if args.reference_body:
        if hp.voc_mode != 'RAW':
            raise ValueError('--reference_body: pass your own MOL loss to voc_train_loop; the CLI only carries F.cross_entropy')
        loss_func = F.cross_entropy

    def at_checkpoint(mod, test_set, step):       # gen_testset (dataset.py:18-43): vocode a few held-out mels next to the checkpoint
        from .dsp import decode_mu_law, label_2_float, save_wav
        k = step // 1000
        batch_str = f'gen_batched_target{hp.voc_target}_overlap{hp.voc_overlap}' if hp.voc_gen_batched else 'gen_NOT_BATCHED'
        for i, (wav, mel) in enumerate(test_set[:hp.voc_gen_at_checkpoint], 1):
            x = np.load(wav)                          # the quantised target (:28-37): decoded and kept next to the generated file, as the reference does
            bits = 16 if hp.voc_mode == 'MOL' else hp.bits
            x = decode_mu_law(x, 2 ** bits, from_labels=True) if (hp.mu_law and hp.voc_mode != 'MOL') else label_2_float(x.astype(np.float64), bits)
            save_wav(x, paths.voc_output / f'{k}k_steps_{i}_target.wav')
            m = torch.from_numpy(np.load(mel).T.astype(np.float32)).unsqueeze(0)
            mod.generate(m, str(paths.voc_output / f'{k}k_steps_{i}_{batch_str}.wav'), hp.voc_gen_batched, hp.voc_target,
                         hp.voc_overlap, hp.mu_law)

    voc_train_loop(paths, model, loss_func, optimizer, WindowLoader(train, hp.voc_batch_size, **kw), test, hp.voc_lr, total,
                   clip_grad_norm=hp.voc_clip_grad_norm, checkpoint_every=hp.voc_checkpoint_every, at_checkpoint=at_checkpoint,
                   report=lambda s: print('\r' + s, end='', flush=True))
    print('\nTraining Complete.')


if __name__ == "__main__":
    main()
