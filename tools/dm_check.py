"""Developer probe (GPU box): secondary dual-softmax model, team kernel vs single-workgroup kernel + timings."""
import sys, time
import numpy as np
import torch

sys.path.insert(0, '.')
from tacotronv2_wavernn_chinese_amd.deepmind import WaveRNN
from tacotronv2_wavernn_chinese_amd.synth import make_dm_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = WaveRNN()
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in make_dm_state_dict(0).items()})
m.to('cuda:0')
rng = np.random.Generator(np.random.PCG64(3))
q = rng.standard_exponential((n, 2, 256)).astype(np.float32)
res = {}
for name, k in (('single', 1), ('team', 2)):
    m.generate(64, noise=q[:64], kernel=k)
    t0 = time.time()
    out, c, f = m.generate(n, noise=q, kernel=k)
    dt = time.time() - t0
    res[name] = (c, f)
    print(f'{name:6s}: {n} samples in {dt * 1e3:.1f} ms -> {dt / n * 1e6:.2f} us/sample, {n / dt / 1e3:.1f} ksamples/s')
bad = np.argwhere((res['single'][0] != res['team'][0]) | (res['single'][1] != res['team'][1]))
print('first mismatch between kernels:', None if bad.size == 0 else int(bad[0][0]), 'of', n)
