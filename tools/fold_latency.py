"""Developer tool: single-utterance latency of generate(batched=True, ...) over the number of folds and the kernel that runs them,
plus where the host time of one call goes (cProfile + wall stamps).  Feeds the cost model behind target='auto' (vocoder.fold_plan).

    python tools/fold_latency.py [--mode RAW|MOL] [--frames 401] [--profile]
"""
from __future__ import annotations

import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='RAW')
    ap.add_argument('--frames', type=int, default=401)
    ap.add_argument('--overlap', type=int, default=550)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--profile', action='store_true')
    ap.add_argument('--folds', default='8,10,16,24,32,40,48,56,64,80,96,128')
    args = ap.parse_args()
    import torch
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN, fold_target
    dev = torch.device('cuda', 0)
    mode = args.mode
    dims = dict(DEFAULT_DIMS)
    if mode == 'MOL':
        dims['bits'] = 9
    sd = make_state_dict(0, mode=mode, variant='peaky' if mode == 'RAW' else 'default', bits=dims['bits'])
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to(dev)
    mels = make_mels(1000, 1, args.frames)
    total = args.frames * 275
    wave_len = (args.frames - 1) * 275
    td = tempfile.mkdtemp()
    path = os.path.join(td, 'o.wav')

    def run(target, kernel, reps=args.reps, **kw):
        m.generate(mels, path, True, target, args.overlap, True, epilogue='device', seed=1, kernel=kernel, **kw)
        torch.cuda.synchronize(dev)
        ts, loops, pros = [], [], []
        for i in range(reps):
            t0 = time.perf_counter()
            m.generate(mels, path, True, target, args.overlap, True, epilogue='device', seed=2 + i, kernel=kernel, **kw)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
            loops.append(m.last_timing['loop_ms'])
            pros.append(m.last_timing['prologue_ms'])
        return np.array(ts) * 1e3, np.array(loops), np.array(pros), dict(m.last_timing)

    print(f'# mode {mode}, frames {args.frames}, total {total} samples, overlap {args.overlap}')
    print('# folds target steps kernel rows | wall ms (min / median) | loop ms | prologue ms | us/step')
    for n in [int(x) for x in args.folds.split(',')]:
        target = fold_target(total, args.overlap, n)
        for kname in (('team2', 'batch_cs') if n <= 16 else ('batch_cs',)):
            try:
                ts, loops, pros, tm = run(target, _cabi.KERNEL_IDS[kname])
            except Exception as ex:
                print(f'{n:4d} {target:6d} {kname}: {ex!r}')
                continue
            print(f'{n:4d} {target:6d} {tm["steps"]:6d} {kname:8s} {tm["rows"]:4d} | {ts.min():7.2f} {np.median(ts):7.2f} | {np.median(loops):7.2f} | {np.median(pros):5.2f} | '
                  f'{np.median(loops) * 1e3 / tm["steps"]:6.3f}', flush=True)
    # reference hp defaults (wavernn_hparams.py:55-57)
    for kname in ('auto', 'team2', 'batch_cs'):
        ts, loops, pros, tm = run(11000, _cabi.KERNEL_IDS[kname])
        print(f'hp-default 11000/550 kernel={kname} ran={_cabi.KERNEL_NAMES[tm["kernel"]]} rows={tm["rows"]} steps={tm["steps"]} | wall {ts.min():.2f} {np.median(ts):.2f} | loop {np.median(loops):.2f}', flush=True)
    if args.profile:
        for tgt in ('auto',):
            pr = cProfile.Profile()
            pr.enable()
            for i in range(5):
                m.generate(mels, path, True, tgt, args.overlap, True, epilogue='device', seed=50 + i)
            pr.disable()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
            print(s.getvalue())


if __name__ == '__main__':
    main()
