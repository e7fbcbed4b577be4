"""Developer tool: the fold legs of bench.py in the context the driver runs them in (after the configs[2] / [4] legs), with per-call times and a
cProfile of the calls -- BENCH_r05 showed 65-75 ms per call where a fresh process measures 48 ms (tools/fold_latency.py).
    python tools/fold_in_bench.py
"""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
print('fresh process:', json.dumps(bench.fold_leg(dev, 'per_xcd')['config']['ms_per_call']))
for cid in (2, 4):
    e = bench.run_config(cid, world=1, rank=0, dev=dev, dry=False, steps=2, warmup=1, frames=bench.T_FRAMES, batch=0, kernel_name='auto', copy_peak=None)
    print('config', cid, e['value'])
for tgt in ('per_xcd', 'auto', 11000, 'per_xcd'):
    pr = cProfile.Profile()
    pr.enable()
    out = bench.fold_leg(dev, tgt)
    pr.disable()
    print(tgt, json.dumps(out['config']))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(12)
    print(s.getvalue()[:3000])
