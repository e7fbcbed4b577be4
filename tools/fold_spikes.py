"""Developer tool: where do the occasional slow calls of the single-utterance fold mode come from?  60 calls of generate(batched=True, target='auto'),
per-call wall time split at the stages of generate(), with every garbage collection logged (gc.callbacks)."""
import gc
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tacotronv2_wavernn_chinese_amd import vocoder as V  # noqa: E402
from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict  # noqa: E402

dev = torch.device('cuda', 0)
sd = make_state_dict(0, variant='peaky')
m = V.WaveRNN(**DEFAULT_DIMS, mode='RAW')
m.verbose = False
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
m.to(dev)
mels = make_mels(1000, 1, 401)
gcs = []
t_gc = [0.0]


def on_gc(phase, info):
    if phase == 'start':
        t_gc[0] = time.perf_counter()
    else:
        gcs.append((info['generation'], (time.perf_counter() - t_gc[0]) * 1e3))


gc.callbacks.append(on_gc)
stamps = {}
real_raw, real_epi, real_save = m.generate_raw, m.epilogue_device, V.save_wav


def wrap(name, fn):
    def f(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        stamps[name] = stamps.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    return f


m.generate_raw = wrap('generate_raw', real_raw)
m.epilogue_device = wrap('epilogue_launch', real_epi)
V.save_wav = wrap('save_wav', real_save)
td = tempfile.mkdtemp()
path = os.path.join(td, 'o.wav')
for mode in ('gc on', 'gc frozen+disabled'):
    if mode != 'gc on':
        gc.collect(); gc.freeze(); gc.disable()
    m.generate(mels, path, True, 'auto', 550, True, epilogue='device', seed=1)
    rows = []
    for i in range(60):
        stamps.clear(); n0 = len(gcs)
        t0 = time.perf_counter()
        m.generate(mels, path, True, 'auto', 550, True, epilogue='device', seed=2 + i)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) * 1e3
        rows.append((dt, m.last_timing['loop_ms'], dict(stamps), gcs[n0:]))
    ts = np.array([r[0] for r in rows])
    print(f'== {mode}: median {np.median(ts):.2f} ms, mean {ts.mean():.2f}, min {ts.min():.2f}, max {ts.max():.2f}; calls > median + 3 ms: {(ts > np.median(ts) + 3).sum()} of {len(ts)}')
    for i, (dt, loop, st, g) in enumerate(rows):
        if dt > np.median(ts) + 3:
            print(f'   call {i}: {dt:.2f} ms, loop kernel {loop:.2f} ms, stages {({k: round(v, 2) for k, v in st.items()})}, collections during the call {[(a, round(b, 2)) for a, b in g]}')
