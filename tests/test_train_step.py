"""N4 of SURVEY.md 8f, the part that makes training possible: ``WaveRNN.training_loss`` = forward() + the training script's loss
with a BACKWARD pass (``wrnn_train_step``, csrc/train.hip), against ``loss.backward()`` of the reference.

  * tests/golden/train_*.npz -- minted from the UNMODIFIED reference module in train() mode by ``python -m oracle.make_golden
    train`` (B=4, T=5 frames = 1 375 steps: the reference's own voc_seq_len): loss, a strided sample of forward()'s output and,
    per parameter, the gradient's L2 norm + 257 strided entries;
  * oracle/torch_ref.py -- the same computation restated in plain torch ops (test infrastructure), pinned to the reference
    module here (``reference`` marker) and to the goldens everywhere; on the GPU box it supplies the FULL gradients (float64).
Tolerances: gradients are sums over 5 500 (batch, step) pairs of fp32 products evaluated in a different order than torch's
(MFMA tiles, two K halves).  The float32 reference digests carry torch's own fp32 summation noise (bias gradients are column sums
with cancellation: two fp32 evaluations differ by a few 1e-4 of the largest entry), so they are asserted at 1e-3 of the largest
entry and 1e-4 on the norms; the float64 restatement is the tight check: 5e-5 of the largest entry on every gradient entry.
"""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden as mg
from oracle import torch_ref as tr
from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = {c['name']: c for c in mg.TRAIN_CASES}


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    c = CASES[name]
    sd = make_state_dict(int(z['weight_seed']), mode=c['mode'], variant=c['variant'], bits=c['bits'])
    x, mels, y = mg.train_inputs(c)
    return z, c, sd, x, mels, y


def _check_against_golden(z, loss, logits, grads, tol):
    assert abs(loss - float(z['loss'])) <= 2e-5 * max(1.0, abs(float(z['loss']))), (loss, float(z['loss']))
    sub = logits[:, ::int(z['sub_stride'])]
    assert np.abs(sub - z['logits_sub']).max() <= 2e-5 * np.abs(z['logits_sub']).max()
    worst = 0.0
    for k in z['keys']:
        k = str(k)
        g = np.asarray(grads[k], np.float32).reshape(-1)
        norm = float(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
        ref_norm = float(z['norm/' + k])
        # the three up-layer FIRs (11/11/23 taps, :75-79) collect their gradient from all B x L x 80 upsampled positions with
        # heavy cancellation: fp32 summation order shows at a few 1e-4 there
        loose = 5.0 if 'up_layers' in k else 1.0
        assert abs(norm - ref_norm) <= loose * 1e-4 * max(ref_norm, 1e-6), (k, norm, ref_norm)
        val, idx = z['val/' + k], z['idx/' + k]
        scale = max(float(np.abs(val).max()), ref_norm / np.sqrt(g.size), 1e-12)
        err = float(np.abs(g[idx] - val).max()) / scale
        worst = max(worst, err)
        assert err <= loose * tol, (k, err)
    return worst


@pytest.mark.parametrize('name', sorted(CASES))
def test_torch_restatement_matches_the_reference_golden(name):
    """CPU, anywhere: oracle/torch_ref.py (float32) reproduces the reference's loss, forward sample and every gradient digest."""
    z, c, sd, x, mels, y = _load(name)
    out = tr.training_step(sd, c['mode'], x, mels, y)
    assert len(z['keys']) == 84 and set(map(str, z['keys'])) == set(out['grads'])
    worst = _check_against_golden(z, out['loss'], out['logits'], out['grads'], 1e-3)
    print(f'\n[train {name}] torch restatement vs reference golden: worst gradient sample error {worst:.2e} (of the largest entry)')


@pytest.mark.reference
@pytest.mark.parametrize('name', ['train_mol_default_b4_t5'])
def test_torch_restatement_equals_the_reference_module(name):
    """Build container: the restatement against the unmodified reference module itself -- FULL gradients, not digests (the MOL
    case: it differs from RAW in the loss only, and the RAW loss is torch's own F.cross_entropy on both sides)."""
    from oracle import ref_harness as rh
    z, c, sd, x, mels, y = _load(name)
    model = rh.build_reference_model(sd, mode=c['mode'], bits=c['bits'])
    ref = rh.reference_train_step(model, x, mels, y)
    out = tr.training_step(sd, c['mode'], x, mels, y)
    assert abs(out['loss'] - ref['loss']) <= 1e-5 * max(1.0, abs(ref['loss']))
    np.testing.assert_allclose(out['logits'], ref['logits'], rtol=0, atol=2e-5 * np.abs(ref['logits']).max())
    assert set(out['grads']) == set(ref['grads'])
    for k, g in ref['grads'].items():
        np.testing.assert_allclose(out['grads'][k], g, rtol=0, atol=2e-4 * max(np.abs(g).max(), 1e-12), err_msg=k)


def _model(c, sd):
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = c['bits']
    m = WaveRNN(**dims, mode=c['mode'])
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.train()
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_training_loss_backward_matches_the_reference(name):
    """MI355X: training_loss(x, mels, y).backward() -- loop layers forward + backward in wrnn_train_step, upsample network through
    autograd -- against (1) the reference golden and (2) the float64 restatement: loss, forward output and ALL 84 gradients."""
    z, c, sd, x, mels, y = _load(name)
    m = _model(c, sd)
    step0 = m.get_step()
    loss, logits = m.training_loss(x, mels, y, return_logits=True)
    loss.backward()
    assert m.get_step() == step0 + 1
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    assert set(grads) == set(map(str, z['keys']))
    worst = _check_against_golden(z, float(loss), logits.cpu().numpy(), grads, 1e-3)
    ref = tr.training_step(sd, c['mode'], x, mels, y, dtype=torch.float64, device='cuda')   # torch float64 on the GPU: the checker
    worst64 = 0.0
    for k, g in ref['grads'].items():
        scale = max(float(np.abs(g).max()), 1e-12)
        err = float(np.abs(grads[k] - g).max()) / scale
        worst64 = max(worst64, err)
        # loop layers (wrnn_train_step): 5e-5.  upsample network: its gradients continue from d_mels_up / d_aux through torch's
        # float32 autograd (21 BatchNorms in training mode, FIRs with heavy cancellation): float32-vs-float64 noise of a few 1e-4
        # End to end the conditioning comes from torch's float32 upsample network on the GPU (MIOpen convolutions, BatchNorm on
        # batch statistics): mels_up / aux differ from the float64 ones at the 1e-6 level and every gradient inherits a few 1e-4
        # of its largest entry (torch's own float32 restatement on this GPU: 3.5e-4 on I.weight, 1.2e-3 on fc2.weight).  The
        # tight check of wrnn_train_step itself is test_loop_gradients_from_float64_conditioning below (3e-6 measured).
        assert err <= 2e-3, (k, err)
    print(f'\n[train {name}] loss {float(loss):.6f} (reference {float(z["loss"]):.6f}); worst gradient error vs the reference digests '
          f'{worst:.2e}, vs the float64 restatement (all {sum(g.size for g in grads.values())} entries) {worst64:.2e}')


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_loop_gradients_from_float64_conditioning(name):
    """wrnn_train_step alone: fed the conditioning of the float64 restatement (rounded to float32), its 16 parameter gradients
    and d_mels_up / d_aux (what flows back into the upsample network) against float64 autograd: within 2e-5 of the largest entry
    (measured: RAW 8e-6, MOL 1.3e-6 -- `mol_grad_kernel` evaluates the discretised likelihood's gradient in double: in fp32, the
    reference's arithmetic, cdf_plus - cdf_min of two sigmoids 1/65535 apart carries ~1e-3 relative noise and torch's own float32
    gradients differ from float64 by 1.1e-4 of the largest entry on I.weight).  Also the g == NULL mode
    (forward + loss only) and the replay of the captured step graphs."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    z, c, sd, x, mels, y = _load(name)
    tol = 2e-5
    m = _model(c, sd)
    ref = tr.training_step(sd, c['mode'], x, mels, y, dtype=torch.float64, device='cuda')
    dev = torch.device('cuda:0')
    mu = torch.from_numpy(ref['mels_up'].astype(np.float32)).to(dev).contiguous()
    au = torch.from_numpy(ref['aux'].astype(np.float32)).to(dev).contiguous()
    xt = torch.from_numpy(x).to(dev)
    yt = torch.from_numpy(y).to(dev).to(torch.int32 if c['mode'] == 'RAW' else torch.float32).contiguous()
    ps = [p.detach().contiguous() for p in m._loop_params()]
    gs = [torch.empty_like(p) for p in ps]
    dm, da = torch.empty_like(mu), torch.empty_like(au)
    loss, loss2 = torch.empty((), device=dev), torch.empty((), device=dev)
    nat = m._native_handle()
    B, L = x.shape
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):   # the second call replays the captured step graphs
        nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], xt.data_ptr(), mu.data_ptr(), au.data_ptr(), yt.data_ptr(),
                       B, L, loss.data_ptr(), 0, dm.data_ptr(), da.data_ptr(), st)
    nat.train_step([p.data_ptr() for p in ps], None, xt.data_ptr(), mu.data_ptr(), au.data_ptr(), yt.data_ptr(), B, L, loss2.data_ptr(), 0, 0, 0, st)
    torch.cuda.synchronize()
    assert abs(float(loss) - ref['loss']) <= 2e-5 * abs(ref['loss']) and float(loss2) == float(loss)
    worst = 0.0
    for got, want, nm in [(dm, ref['d_mels_up'], 'd_mels_up'), (da, ref['d_aux'], 'd_aux')] + \
            [(g, ref['grads'][k], k) for g, k in zip(gs, _cabi.LOOP_PARAM_KEYS)]:
        err = float(np.abs(got.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-12))
        worst = max(worst, err)
        assert err <= tol, (nm, err)
    print(f'\n[train {name}] wrnn_train_step vs float64 autograd on the same conditioning: worst error {worst:.2e} of the largest entry')


@pytest.mark.gpu
def test_a_few_optimizer_steps_at_the_reference_batch_size():
    """The reference's own training shape (wavernn_hparams.py: voc_batch_size 32, voc_seq_len 5 hops): Adam + clip_grad_norm_ as in
    wavernn_train.py:122-128; the loss of a fixed batch must fall, and the step time is printed (samples of audio per second)."""
    import time
    c = CASES['train_raw_peaky_b4_t5']
    sd = make_state_dict(0, mode='RAW', variant='default', bits=10)
    m = _model(dict(c, variant='default'), sd)
    B, T = 32, 5
    rng = np.random.Generator(np.random.PCG64(3))
    lab = rng.integers(0, 1024, size=(B, T * 275 + 1))
    lab[:, 1:] = (lab[:, :1] + np.cumsum(rng.integers(-3, 4, size=(B, T * 275)), axis=1)) % 1024      # a learnable random walk
    x = (2.0 * lab[:, :-1] / 1023.0 - 1.0).astype(np.float32)
    y = lab[:, 1:]
    mels = rng.random((B, 80, T + 4), dtype=np.float32)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses, t0 = [], None
    for it in range(8):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = m.training_loss(x, mels, y)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 4)
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f'\n[train] B=32, L=1375: {dt * 1e3:.1f} ms per iteration (forward + backward + Adam) = {B * T * 275 / dt / 1e3:.0f} ksamples/s; '
          f'loss {losses[0]:.4f} -> {losses[-1]:.4f}')
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_train_step_on_non_default_dims_and_ragged_batch_sizes(mode):
    """The generic instantiations (`gru_*_step_kernel<0>`: any rnn_dims % 16 == 0) and a batch that is not a multiple of the 32-row
    tile: rnn 256 / fc 384 / 40 mels / aux 24, B = 5 and B = 35, against float64 autograd on the same conditioning."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dims = dict(rnn_dims=256, fc_dims=384, bits=8, pad=2, upsample_factors=(4, 4, 8), feat_dims=40, compute_dims=64,
                res_out_dims=96, res_blocks=2, hop_length=128, sample_rate=16000)
    sd = make_state_dict(7, mode=mode, variant='default', **dims)
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to('cuda:0')
    m.train()
    dev = torch.device('cuda:0')
    for B, T in ((5, 2), (35, 1)):
        L = T * 128
        rng = np.random.Generator(np.random.PCG64(B))
        mels = rng.random((B, 40, T + 4), dtype=np.float32)
        if mode == 'RAW':
            lab = rng.integers(0, 256, size=(B, L + 1))
            x, y = (2.0 * lab[:, :-1] / 255.0 - 1.0).astype(np.float32), lab[:, 1:].astype(np.int64)
        else:
            sig = rng.uniform(-1, 1, size=(B, L + 1)).astype(np.float32)
            x, y = sig[:, :-1].copy(), sig[:, 1:].copy()
        sd64 = {k: torch.as_tensor(np.asarray(v)).to(dev, torch.float64).requires_grad_(torch.as_tensor(np.asarray(v)).is_floating_point()
                                                                                          and not k.endswith(('running_mean', 'running_var')))
                if torch.as_tensor(np.asarray(v)).is_floating_point() else torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
        mu64, au64 = tr.upsample(sd64, torch.from_numpy(mels).to(dev, torch.float64), upsample_factors=(4, 4, 8), pad=2, training=True)
        mu64.retain_grad()
        au64.retain_grad()
        loss64 = tr.loss_of(mode, tr.loop_forward(sd64, torch.from_numpy(x).to(dev, torch.float64), mu64, au64), torch.from_numpy(y).to(dev))
        loss64.backward()
        mu = mu64.detach().float().contiguous()
        au = au64.detach().float().contiguous()
        xt = torch.from_numpy(x).to(dev)
        yt = torch.from_numpy(y).to(dev).to(torch.int32 if mode == 'RAW' else torch.float32).contiguous()
        ps = [p.detach().contiguous() for p in m._loop_params()]
        gs = [torch.empty_like(p) for p in ps]
        dm, da = torch.empty_like(mu), torch.empty_like(au)
        loss = torch.empty((), device=dev)
        m._native_handle().train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], xt.data_ptr(), mu.data_ptr(), au.data_ptr(),
                                      yt.data_ptr(), B, L, loss.data_ptr(), 0, dm.data_ptr(), da.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(loss64)) <= 2e-5 * abs(float(loss64)), (B, float(loss), float(loss64))
        # Random initial weights put a few of the B*L*(fc1 + fc2) ReLU pre-activations within float32 rounding of zero: float32 and
        # float64 then disagree about ONE unit's mask at ONE (batch, step) pair, which moves that pair's row of d_mels_up / d_aux by
        # a few per cent of the tensor's largest entry, and that unit's row of a weight gradient by a few 1e-4 (found with B = 35, T = 1:
        # the fc2 mask of row (10, 93)).  So: 99 % of the entries within 5e-5 of the largest one, no entry off by more than 0.1 of it.
        for got, want, nm in [(dm, mu64.grad, 'd_mels_up'), (da, au64.grad, 'd_aux')] + [(g, sd64[k].grad, k) for g, k in zip(gs, _cabi.LOOP_PARAM_KEYS)]:
            w = want.detach().cpu().numpy()
            err = np.abs(got.cpu().numpy() - w).reshape(-1) / max(np.abs(w).max(), 1e-12)
            assert np.quantile(err, 0.99) <= 5e-5 and err.max() <= 0.1, (mode, B, nm, float(np.quantile(err, 0.99)), float(err.max()))


@pytest.mark.gpu
@pytest.mark.parametrize('B', [3, 32, 61, 70])
def test_team_recurrence_kernels_equal_the_step_kernels(B):
    """The persistent XCD-team kernels of the two GRU recurrences (train_team.hip; 4 or 8 rows per team, K-split backward with a
    reduce-scatter) against the per-step kernels (wrnn_train_force_step_kernels): same loss, same fc3 outputs, gradients equal to
    fp32 summation-order noise.  B = 3: one partial quad; 32: 8 teams x 4 rows; 61: 8 rows per team with a ragged last batch; 70: 9 batches, team 0 runs two of them back to back."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    c = CASES['train_raw_peaky_b4_t5']
    sd = make_state_dict(0, mode='RAW', variant='peaky', bits=10)
    m = _model(c, sd)
    dev = torch.device('cuda:0')
    T = 2
    L = T * 275
    rng = np.random.Generator(np.random.PCG64(100 + B))
    lab = rng.integers(0, 1024, size=(B, L + 1))
    x = torch.from_numpy((2.0 * lab[:, :-1] / 1023.0 - 1.0).astype(np.float32)).to(dev)
    y = torch.from_numpy(lab[:, 1:].astype(np.int32)).to(dev)
    mu = torch.from_numpy(rng.random((B, L, 80), dtype=np.float32)).to(dev)
    au = torch.from_numpy(rng.standard_normal((B, L, 128)).astype(np.float32)).to(dev)
    ps = [p.detach().contiguous() for p in m._loop_params()]
    nat = m._native_handle()
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for mode in ('steps', 'team'):
        nat.train_force_step_kernels(mode == 'steps')
        gs = [torch.zeros_like(p) for p in ps]
        dm, da = torch.zeros_like(mu), torch.zeros_like(au)
        loss = torch.zeros((), device=dev)
        logits = torch.zeros((B, L, 1024), device=dev)
        for _ in range(2):
            nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], x.data_ptr(), mu.data_ptr(), au.data_ptr(), y.data_ptr(), B, L,
                           loss.data_ptr(), logits.data_ptr(), dm.data_ptr(), da.data_ptr(), st)
            nat.sync_status(st)
        res[mode] = (float(loss), logits.cpu().numpy(), [g.cpu().numpy() for g in gs], dm.cpu().numpy(), da.cpu().numpy())
    nat.train_force_step_kernels(False)
    a, b = res['steps'], res['team']
    assert abs(a[0] - b[0]) <= 2e-6 * abs(a[0])
    assert np.abs(a[1] - b[1]).max() <= 2e-5 * np.abs(a[1]).max()
    worst = 0.0
    for ga, gb, k in zip(a[2], b[2], _cabi.LOOP_PARAM_KEYS):
        e = np.abs(ga - gb).reshape(-1) / max(np.abs(ga).max(), 1e-12)
        worst = max(worst, float(e.max()))
        assert float(np.quantile(e, 0.99)) <= 1e-4 and float(e.max()) <= 1e-3, (k, float(np.quantile(e, 0.99)), float(e.max()))
    # d_mels_up / d_aux are per (row, step): two fp32 forwards differ by ~1e-7 in the fc1 / fc2 pre-activations, so a ReLU unit that sits
    # within that of zero takes the other branch in one of them, and that (row, step) -- plus the few steps before it, through the GRU
    # carries -- gets a visibly different gradient (with the "peaky" fc3 x 8 weights: up to 5e-2 of the largest entry).  Measured with
    # tools/diag_team_batches.py (profiles/r03_team_vs_steps_diag.txt): B = 70: 20 of 38 500 (row, step) pairs in 2 rows, none of them in
    # the team's second batch; float64 autograd disagrees with BOTH fp32 variants in the same way (B = 61: the same 28 pairs in either).
    # So: everything else equal to 1e-5, and at most 0.2 % of the pairs touched by a flip.
    for ga, gb, k in [(a[3], b[3], 'd_mels_up'), (a[4], b[4], 'd_aux')]:
        e = np.abs(ga - gb).max(axis=2) / max(np.abs(ga).max(), 1e-12)            # (row, step)
        flipped = int((e > 1e-4).sum())
        assert float(np.quantile(e, 0.99)) <= 1e-5 and flipped <= 0.002 * e.size, (k, float(np.quantile(e, 0.99)), flipped, e.size)
    print(f'\n[train] team vs step recurrence kernels, B={B}: worst gradient difference {worst:.2e} of the largest entry')


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_the_reference_training_loop_body_runs_unchanged(name):
    """`y_hat = model(x, m)` in train() mode is differentiable (wrnn_train_forward / wrnn_train_backward behind autograd), so the loop
    body of wavernn_train.py:103-122 -- its reshaping, torch's own F.cross_entropy / a torch restatement of the MOL loss,
    loss.backward() -- runs as written.  Same loss and gradients as the reference golden; y_hat equals the reference's train-mode
    forward output; in eval() / no_grad the inference-kernel pass is used and gives the eval-mode output."""
    import torch.nn.functional as F
    z, c, sd, x, mels, y = _load(name)
    m = _model(c, sd)
    dev = torch.device('cuda:0')
    xt, mt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(mels).to(dev), torch.from_numpy(y).to(dev)
    y_hat = m(xt, mt)
    assert y_hat.requires_grad and tuple(y_hat.shape) == (c['B'], c['T'] * 275, m.n_classes)
    if m.mode == 'RAW':                                      # wavernn_train.py:112-121
        loss = F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), yt.long().unsqueeze(-1))
    else:
        loss = tr.discretized_mix_logistic_loss(y_hat, yt.float())
    m.zero_grad()
    loss.backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters() if p.grad is not None}
    worst = _check_against_golden(z, float(loss.detach()), y_hat.detach().cpu().numpy(), grads, 1e-3)
    # the fused path gives the same thing
    m.zero_grad()
    loss2 = m.training_loss(x, mels, y)
    loss2.backward()
    assert abs(float(loss2) - float(loss.detach())) <= 2e-6 * abs(float(loss2))
    for k, p in m.named_parameters():
        g2 = p.grad.detach().cpu().numpy()
        tol = 2e-3 if c['mode'] == 'MOL' else 2e-5           # MOL: torch's fp32 loss gradient vs the double-precision mol_grad_kernel
        assert np.abs(g2 - grads[k]).max() <= tol * max(np.abs(g2).max(), 1e-12), k
    m.eval()
    with torch.no_grad():
        ev = m(xt, mt)
    assert not ev.requires_grad and tuple(ev.shape) == tuple(y_hat.shape)
    print(f'\n[train {name}] unchanged loop body: loss {float(loss.detach()):.6f}, worst gradient sample error vs the reference digests {worst:.2e}')


@pytest.mark.gpu
def test_split_forward_backward_survive_a_batch_size_change_on_the_step_kernels():
    """Round-3 advisor finding: on the per-step-kernel path (hipGraph replay) a forward-only call re-captured only the forward graphs
    when (B, L) changed, and the following backward replayed the PREVIOUS problem's graph (stale grid, stale workspace pointers).
    `y_hat = model(x, m); loss.backward()` at B = 6 and then at B = 3 -- the reference loop body when the loader's last batch is
    short -- against the same B = 3 pass on the team kernels."""
    import torch.nn.functional as F
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    c = CASES['train_raw_peaky_b4_t5']
    sd = make_state_dict(0, mode='RAW', variant='peaky', bits=10)
    m = _model(c, sd)
    dev = torch.device('cuda:0')
    nat = m._native_handle()
    T, hop = 2, 275

    def one_pass(B, seed):
        rng = np.random.Generator(np.random.PCG64(seed))
        lab = rng.integers(0, 1024, size=(B, T * hop + 1))
        xt = torch.from_numpy((2.0 * lab[:, :-1] / 1023.0 - 1.0).astype(np.float32)).to(dev)
        yt = torch.from_numpy(lab[:, 1:].astype(np.int64)).to(dev)
        mt = torch.from_numpy(make_mels(seed, B, T + 4)).to(dev)
        y_hat = m(xt, mt)
        loss = F.cross_entropy(y_hat.transpose(1, 2).unsqueeze(-1), yt.unsqueeze(-1))
        m.zero_grad()
        loss.backward()
        return float(loss.detach()), {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}

    m.train()
    try:
        nat.train_force_step_kernels(True)
        one_pass(6, 11)                       # captures the graphs of the (6, L) problem
        loss_s, g_s = one_pass(3, 12)         # forward re-captures; the backward must too
        nat.train_force_step_kernels(False)
        loss_t, g_t = one_pass(3, 12)         # the same pass on the persistent team kernels
    finally:
        nat.train_force_step_kernels(False)
    assert abs(loss_s - loss_t) <= 2e-6 * abs(loss_t)
    for k in g_t:
        e = np.abs(g_s[k] - g_t[k]).max() / max(np.abs(g_t[k]).max(), 1e-12)
        assert e <= 2e-3, (k, float(e))


@pytest.mark.gpu
def test_backward_after_another_training_call_fails_loudly():
    """The split pass keeps its activations in the handle's single workspace: `y1 = model(x1, m1); y2 = model(x2, m2); y1.sum().backward()`
    would differentiate the SECOND pass.  The forward is stamped; the stale backward raises (round-3 advisor finding)."""
    from tacotronv2_wavernn_chinese_amd.synth import make_mels
    c = CASES['train_raw_peaky_b4_t5']
    sd = make_state_dict(0, mode='RAW', variant='peaky', bits=10)
    m = _model(c, sd)
    m.train()
    dev = torch.device('cuda:0')
    rng = np.random.Generator(np.random.PCG64(5))
    xs = [torch.from_numpy(rng.uniform(-1, 1, (3, 2 * 275)).astype(np.float32)).to(dev) for _ in range(2)]
    ms = [torch.from_numpy(make_mels(70 + i, 3, 2 + 4)).to(dev) for i in range(2)]
    y1 = m(xs[0], ms[0])
    y2 = m(xs[1], ms[1])
    with pytest.raises(RuntimeError, match='another training call'):
        y1.sum().backward()
    m.zero_grad()
    y2.sum().backward()                      # the latest pass is still differentiable
    assert all(p.grad is not None for k, p in m.named_parameters() if not k.startswith('upsample'))
