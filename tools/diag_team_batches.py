"""Diagnostic (GPU box): where do the team recurrence kernels and the per-step kernels of wrnn_train_step differ, and which one agrees
with float64 autograd?  `python tools/diag_team_batches.py 70 72 130` -> per B: positions (row, step) of d_mels_up / d_aux mismatches,
team-vs-team determinism, and both variants against oracle/torch_ref.py in float64."""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from oracle import torch_ref as tr
from tacotronv2_wavernn_chinese_amd import _cabi
from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_state_dict
from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN


def run(B, T=2, variant='peaky'):
    dev = torch.device('cuda:0')
    sd = make_state_dict(0, mode='RAW', variant=variant, bits=10)
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to(dev).train()
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
    for tag, steps in (('steps', True), ('team', False), ('team2', False), ('steps2', True)):
        nat.train_force_step_kernels(steps)
        gs = [torch.zeros_like(p) for p in ps]
        dm, da = torch.zeros_like(mu), torch.zeros_like(au)
        loss = torch.zeros((), device=dev)
        for _ in range(2):
            nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], x.data_ptr(), mu.data_ptr(), au.data_ptr(), y.data_ptr(), B, L,
                           loss.data_ptr(), 0, dm.data_ptr(), da.data_ptr(), st)
            nat.sync_status(st)
        res[tag] = dict(loss=float(loss), g={k: g.cpu().numpy() for k, g in zip(_cabi.LOOP_PARAM_KEYS, gs)}, dm=dm.cpu().numpy(), da=da.cpu().numpy())
    nat.train_force_step_kernels(False)
    # float64 autograd of the loop layers on the same conditioning
    sd64 = {k: torch.as_tensor(np.asarray(v)).to(dev, torch.float64).requires_grad_(True) for k, v in sd.items()
            if np.asarray(v).dtype.kind == 'f' and k.split('.')[0] in ('I', 'rnn1', 'rnn2', 'fc1', 'fc2', 'fc3')}
    mu64, au64 = mu.double().requires_grad_(True), au.double().requires_grad_(True)
    loss64 = tr.loss_of('RAW', tr.loop_forward(sd64, x.double(), mu64, au64), y)
    loss64.backward()
    ref = dict(loss=float(loss64), dm=mu64.grad.cpu().numpy(), da=au64.grad.cpu().numpy(), g={k: sd64[k].grad.cpu().numpy() for k in _cabi.LOOP_PARAM_KEYS})
    print(f'== B={B} L={L}: loss steps {res["steps"]["loss"]:.6f} team {res["team"]["loss"]:.6f} f64 {ref["loss"]:.6f}')
    for a, b in (('steps', 'team'), ('team', 'team2'), ('steps', 'steps2'), ('steps', 'ref'), ('team', 'ref')):
        ra, rb = res[a], (ref if b == 'ref' else res[b])
        for nm in ('dm', 'da'):
            d = np.abs(ra[nm] - rb[nm]) / max(np.abs(rb[nm]).max(), 1e-30)
            bad = np.argwhere(d.max(axis=2) > 1e-4)
            rows = sorted(set(bad[:, 0].tolist()))
            print(f'  {a:6s} vs {b:6s} {nm}: max {d.max():.2e}, p99 {np.quantile(d, 0.99):.1e}, (row, step) pairs above 1e-4: {len(bad)}; rows {rows[:20]}',
                  f'steps of the first bad row: {bad[bad[:, 0] == rows[0]][:, 1][:12].tolist()} .. {bad[bad[:, 0] == rows[0]][:, 1][-4:].tolist()}' if rows else '')
        worst = max((float(np.abs(ra['g'][k] - rb['g'][k]).max() / max(np.abs(rb['g'][k]).max(), 1e-30)), k) for k in _cabi.LOOP_PARAM_KEYS)
        print(f'  {a:6s} vs {b:6s} parameter gradients: worst {worst[0]:.2e} ({worst[1]})')


if __name__ == '__main__':
    for B in [int(v) for v in sys.argv[1:]] or [70]:
        run(B)
