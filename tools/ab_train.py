"""Developer A/B harness (GPU box) for the training step: time wrnn_train_step alone (HIP events, no torch ops in the timed region) on
one or more builds of the library.  `python tools/ab_train.py [B] [T] -- libA.so libB.so ...` (each library in its own process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib, B, T, iters=12):
    import numpy as np
    import torch
    from tacotronv2_wavernn_chinese_amd import _cabi
    if lib != 'product':
        _cabi.LIB_PATH = os.path.abspath(lib)
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.verbose = False
    m.to(dev).train()
    L = T * 275
    rng = np.random.Generator(np.random.PCG64(1))
    x = torch.from_numpy(rng.uniform(-1, 1, (B, L)).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.integers(0, 1024, (B, L)).astype(np.int32)).to(dev)
    mu = torch.from_numpy(rng.random((B, L, 80), dtype=np.float32)).to(dev)
    au = torch.from_numpy(rng.standard_normal((B, L, 128)).astype(np.float32)).to(dev)
    ps = [p.detach().contiguous() for p in m._loop_params()]
    gs = [torch.zeros_like(p) for p in ps]
    dm, da = torch.zeros_like(mu), torch.zeros_like(au)
    loss = torch.zeros((), device=dev)
    nat = m._native_handle()
    st = torch.cuda.current_stream(dev).cuda_stream

    def call():
        nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], x.data_ptr(), mu.data_ptr(), au.data_ptr(), y.data_ptr(), B, L,
                       loss.data_ptr(), 0, dm.data_ptr(), da.data_ptr(), st)
    for _ in range(3):
        call()
    nat.sync_status(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    nat.sync_status(st)
    ms = e0.elapsed_time(e1) / iters
    chk = float(sum(float(g.double().abs().sum()) for g in gs))
    print(f'{os.path.basename(lib):28s} B={B} L={L}: wrnn_train_step {ms:.2f} ms = {B * L / ms:.0f} ksamples/s; loss {float(loss):.6f}, sum|grad| {chk:.6e}', flush=True)


if __name__ == '__main__':
    if '--one' in sys.argv:
        i = sys.argv.index('--one')
        one(sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3]))
    else:
        cut = sys.argv.index('--') if '--' in sys.argv else len(sys.argv)
        nums = [int(v) for v in sys.argv[1:cut]]
        B, T = (nums + [32, 5])[:2] if len(nums) < 2 else nums[:2]
        for lib in sys.argv[cut + 1:] or ['product']:
            subprocess.run([sys.executable, __file__, '--one', lib, str(B), str(T)], check=False)
