"""TEST INFRASTRUCTURE -- mints tests/golden/*.npz from the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [long | forward | train | dm]

Every fixture stores only seeds + the reference's outputs: weights, mels and
the sampling noise are rebuilt from the seeds on whatever box replays them
(``tacotronv2_wavernn_chinese_amd.synth`` is numpy-PCG64; the noise is the
torch CPU generator stream, replayed by ``noise_from_seed`` below and guarded
by a checksum stored in the fixture so an RNG drift is detected, not silently
compared).
"""
from __future__ import annotations

import os
import sys

import numpy as np

from oracle.noise import noise_checksum
from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')

# name, mode, bits, variant, B, T, batched, target, overlap
CASES = [
    dict(name='raw_default_b1_t24', mode='RAW', bits=10, variant='default', B=1, T=24, batched=False),
    dict(name='raw_peaky_b1_t24', mode='RAW', bits=10, variant='peaky', B=1, T=24, batched=False),
    dict(name='raw_peaky_b3_t21', mode='RAW', bits=10, variant='peaky', B=3, T=21, batched=False),
    dict(name='raw_peaky_fold_t30', mode='RAW', bits=10, variant='peaky', B=1, T=30, batched=True,
         target=2000, overlap=200),
    dict(name='mol_default_b1_t24', mode='MOL', bits=9, variant='default', B=1, T=24, batched=False),
    dict(name='mol_default_b2_t21', mode='MOL', bits=9, variant='default', B=2, T=21, batched=False),
]
# BASELINE-size cases (`python -m oracle.make_golden long`): configs[1] itself (B=1, T=401: 110 275 free-running
# steps = 14 default segments of the shipped kernel) and a B=8 clip long enough for several natural segments
# per row.  Only labels (+ the B=1 wav) are stored: the conditioning slices of the short cases already pin A2-A5.
LONG_CASES = [
    dict(name='raw_peaky_b1_t401', mode='RAW', bits=10, variant='peaky', B=1, T=401, batched=False, slim=True),
    dict(name='raw_peaky_b8_t60', mode='RAW', bits=10, variant='peaky', B=8, T=60, batched=False, slim=True),
]
# Teacher-forced forward() + loss (`python -m oracle.make_golden forward`): N4 of SURVEY.md 8f
FORWARD_CASES = [
    dict(name='fwd_raw_peaky_b2_t6', mode='RAW', bits=10, variant='peaky', B=2, T=6),
    dict(name='fwd_mol_default_b2_t6', mode='MOL', bits=9, variant='default', B=2, T=6),
    # wider: 4 rows x 20 frames (5 500 steps per row: several frame changes per row, more rows than a row quad of the batch kernel)
    dict(name='fwd_raw_peaky_b4_t20', mode='RAW', bits=10, variant='peaky', B=4, T=20, sub=97),
    dict(name='fwd_mol_default_b4_t20', mode='MOL', bits=9, variant='default', B=4, T=20, sub=23),
]
# One training iteration (`python -m oracle.make_golden train`): forward in train() mode (BatchNorm on batch statistics), loss,
# loss.backward() -- the loss and, per parameter, the gradient's L2 norm + a strided sample of it (the full gradients are 17 MB)
TRAIN_CASES = [
    dict(name='train_raw_peaky_b4_t5', mode='RAW', bits=10, variant='peaky', B=4, T=5),
    dict(name='train_mol_default_b4_t5', mode='MOL', bits=9, variant='default', B=4, T=5),
]
# The secondary dual-softmax model (`python -m oracle.make_golden dm`): SURVEY.md 8a A12, deepmind_version.py:75-165
DM_CASES = [dict(name='dm_h896_s2000', hidden=896, steps=2000)]
TRAIN_GRAD_SAMPLES = 257
WEIGHT_SEED, MEL_SEED, NOISE_SEED = 0, 1234, 42


def train_inputs(c):
    """Seeded (x, mels, y) of a training fixture: rebuilt identically wherever the fixture is replayed."""
    B, T, pad, hop = c['B'], c['T'], 2, 275
    L = T * hop
    mels = make_mels(MEL_SEED, B, T + 2 * pad)
    rng = np.random.Generator(np.random.PCG64(NOISE_SEED + 1))
    if c['mode'] == 'RAW':
        lab = rng.integers(0, 2 ** c['bits'], size=(B, L + 1))
        x = (2.0 * lab[:, :-1] / (2 ** c['bits'] - 1.0) - 1.0).astype(np.float32)     # the dataset's x / y: a signal and its shift
        y = lab[:, 1:].astype(np.int64)
    else:
        sig = rng.uniform(-1.0, 1.0, size=(B, L + 1)).astype(np.float32)
        sig[0, 1:8] = [-1.0, -0.9995, 0.9995, 1.0, 0.0, 0.5, -0.5]                      # the edge branches of the discretised likelihood
        x, y = sig[:, :-1].copy(), sig[:, 1:].copy()
    return x, mels, y


def grad_digest(g: np.ndarray) -> dict:
    flat = np.asarray(g, np.float32).reshape(-1)
    idx = np.linspace(0, flat.size - 1, min(TRAIN_GRAD_SAMPLES, flat.size)).astype(np.int64)
    return dict(norm=np.float64(np.sqrt(np.sum(flat.astype(np.float64) ** 2))), idx=idx, val=flat[idx])


def mint_train():
    from oracle import ref_harness as rh
    for c in TRAIN_CASES:
        sd = make_state_dict(WEIGHT_SEED, mode=c['mode'], variant=c['variant'], bits=c['bits'])
        model = rh.build_reference_model(sd, mode=c['mode'], bits=c['bits'])
        x, mels, y = train_inputs(c)
        out = rh.reference_train_step(model, x, mels, y)
        fix = dict(mode=c['mode'], bits=c['bits'], variant=c['variant'], B=c['B'], T=c['T'], weight_seed=WEIGHT_SEED, mel_seed=MEL_SEED,
                   xy_seed=NOISE_SEED + 1, loss=np.float64(out['loss']), logits_sub=out['logits'][:, ::97].astype(np.float32), sub_stride=97,
                   keys=np.array(sorted(out['grads'])))
        for k, g in out['grads'].items():
            dg = grad_digest(g)
            fix['norm/' + k], fix['idx/' + k], fix['val/' + k] = dg['norm'], dg['idx'], dg['val']
        path = os.path.join(GOLDEN_DIR, c['name'] + '.npz')
        np.savez_compressed(path, **fix)
        print(f"{c['name']}: loss {out['loss']:.6f}, {len(out['grads'])} gradients -> {os.path.getsize(path) / 1024:.0f} KiB")



def mint_forward():
    from oracle import ref_harness as rh
    for c in FORWARD_CASES:
        sd = make_state_dict(WEIGHT_SEED, mode=c['mode'], variant=c['variant'], bits=c['bits'])
        model = rh.build_reference_model(sd, mode=c['mode'], bits=c['bits'])
        B, T, pad, hop = c['B'], c['T'], 2, 275
        L = T * hop
        mels = make_mels(MEL_SEED, B, T + 2 * pad)          # the padded window the training collate hands over
        rng = np.random.Generator(np.random.PCG64(NOISE_SEED))
        if c['mode'] == 'RAW':
            y = rng.integers(0, 1024, size=(B, L)).astype(np.int64)
            x = (2.0 * rng.integers(0, 1024, size=(B, L)) / 1023.0 - 1.0).astype(np.float32)
        else:
            y = rng.uniform(-1.0, 1.0, size=(B, L)).astype(np.float32)
            y[0, :7] = [-1.0, -0.9995, 0.9995, 1.0, 0.0, 0.5, -0.5]   # the edge branches of the discretised likelihood
            x = rng.uniform(-1.0, 1.0, size=(B, L)).astype(np.float32)
        out = rh.reference_forward(model, x, mels, y)
        stride = c.get('sub', 23)
        sub = slice(0, L, stride)
        import torch
        loss_sub = out['loss_of'](torch.from_numpy(np.ascontiguousarray(out['logits'][:, sub])), y[:, sub])
        fix = dict(mode=c['mode'], bits=c['bits'], variant=c['variant'], B=B, T=T, weight_seed=WEIGHT_SEED, mel_seed=MEL_SEED,
                   xy_seed=NOISE_SEED, x=x, y=y, logits_sub=out['logits'][:, sub].astype(np.float32), sub_stride=stride,
                   loss=np.float64(out['loss']), loss_sub=np.float64(loss_sub))
        path = os.path.join(GOLDEN_DIR, c['name'] + '.npz')
        np.savez_compressed(path, **fix)
        print(f"{c['name']}: logits {out['logits'].shape} loss {out['loss']:.6f} loss_sub {loss_sub:.6f} -> {os.path.getsize(path) / 1024:.0f} KiB")


def mint_dm():
    """deepmind_version.WaveRNN.generate(seq_len) of the unmodified reference on seeded weights: the 16-bit output and the
    coarse / fine class indices of every step.  Stored with the seeds only; tests/test_deepmind.py replays the Exp(1) draws
    (oracle.noise.dm_noise_from_seed) and checks them against the checksum kept here."""
    from oracle import ref_harness as rh
    from oracle.noise import dm_noise_from_seed
    from tacotronv2_wavernn_chinese_amd.synth import make_dm_state_dict
    for c in DM_CASES:
        sd = make_dm_state_dict(WEIGHT_SEED, hidden_size=c['hidden'])
        out = rh.reference_dm_generate(sd, c['steps'], NOISE_SEED)
        q = dm_noise_from_seed(NOISE_SEED, c['steps'])
        path = os.path.join(GOLDEN_DIR, c['name'] + '.npz')
        np.savez_compressed(path, weight_seed=WEIGHT_SEED, noise_seed=NOISE_SEED, steps=c['steps'],
                            coarse=out['coarse'].astype(np.int16), fine=out['fine'].astype(np.int16), output=out['output'].astype(np.int32),
                            noise_checksum=noise_checksum({'q': q}))
        print(f"{c['name']}: {c['steps']} steps -> {os.path.getsize(path) / 1024:.0f} KiB")


def main() -> int:
    from oracle import ref_harness as rh
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'dm':
        mint_dm()
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == 'forward':
        mint_forward()
        return 0
    if len(sys.argv) > 1 and sys.argv[1] == 'train':
        mint_train()
        return 0
    cases = LONG_CASES if (len(sys.argv) > 1 and sys.argv[1] == 'long') else CASES
    for c in cases:
        sd = make_state_dict(WEIGHT_SEED, mode=c['mode'], variant=c['variant'], bits=c['bits'])
        model = rh.build_reference_model(sd, mode=c['mode'], bits=c['bits'])
        mels = make_mels(MEL_SEED, c['B'], c['T'])
        target, overlap = c.get('target', 11000), c.get('overlap', 550)
        out = rh.reference_generate(model, mels, seed=NOISE_SEED, batched=c['batched'], target=target,
                                    overlap=overlap, mu_law=True)
        key = 'labels' if c['mode'] == 'RAW' else 'samples'
        L, rows = out[key].shape
        noise = rh.replay_noise(NOISE_SEED, c['mode'], L, rows, n_classes=1024)
        up, aux = rh.reference_upsample(model, mels)
        fix = dict(
            mode=c['mode'], bits=c['bits'], variant=c['variant'], B=c['B'], T=c['T'],
            batched=c['batched'], target=target, overlap=overlap,
            weight_seed=WEIGHT_SEED, mel_seed=MEL_SEED, noise_seed=NOISE_SEED,
            wav=out['wav'], noise_checksum=noise_checksum(noise),
            up_head=up[:, :320], up_tail=up[:, -320:], up_stride=up[:, ::41],
            aux_frames=aux[:, ::275],
        )
        if c.get('slim'):
            for k in ('up_head', 'up_tail', 'up_stride', 'aux_frames'):
                del fix[k]
            fix['wav'] = out['wav'].astype(np.float32) if c['B'] == 1 else np.zeros(0, np.float32)
            fix['ref_seconds'] = out['seconds']
        if c['mode'] == 'RAW':
            fix['labels'] = out['labels'].astype(np.int16)
        else:
            fix['samples'] = out['samples']
        path = os.path.join(GOLDEN_DIR, c['name'] + '.npz')
        np.savez_compressed(path, **fix)
        print(f"{c['name']}: L={L} rows={rows} wav={out['wav'].shape} ref {out['seconds']:.1f}s "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")
    return 0


if __name__ == '__main__':
    sys.exit(main())
