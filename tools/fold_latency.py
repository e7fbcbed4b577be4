"""Developer probe (GPU box): single-utterance latency, unbatched vs the reference's fold mode vs target='auto'."""
import sys, time
import numpy as np
import torch

sys.path.insert(0, '.')
from tools.quick_check import model
from tacotronv2_wavernn_chinese_amd.synth import make_mels

T = int(sys.argv[1]) if len(sys.argv) > 1 else 401
m = model('RAW')
mels = make_mels(3, 1, T)
audio_s = (T - 1) * 275 / 22050
for name, batched, target in (('unbatched', False, 11000), ('folds target=11000', True, 11000), ("folds target='auto'", True, 'auto')):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        wav = m.generate(mels, '/tmp/o.wav', batched, target, 550, True, epilogue='device', seed=5)
        dt = time.time() - t0
    tm = m.last_timing
    print(f'{name:22s}: rows {tm["rows"]:2d} x {tm["steps"]:6d} steps, loop {tm["loop_ms"]:7.2f} ms, prologue {tm["prologue_ms"]:.2f} ms, '
          f'generate() wall {dt * 1e3:7.1f} ms -> {audio_s / dt:5.1f}x real time ({audio_s:.2f} s of audio)')
