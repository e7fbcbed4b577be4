"""TEST INFRASTRUCTURE -- times the UNMODIFIED reference ``WaveRNN.generate()`` on the host cores.

BASELINE.md section 3 / SURVEY.md 8d: the reference's own CPU path (``wavernn_gen.py`` forces the device to CPU,
``/root/reference/wavernn_gen.py:93,126``) on BASELINE configs[0] (one 80x200 synthetic mel, RAW 10-bit, batch 1) and on
the configs[1] clip (80x401), with all cores and with one thread.  Needs ``/root/reference``, so it runs in the
build container only; ``bench.py`` reads the JSON this writes (``profiles/cpu_reference_container.json``) and
reports it as ``cpu_reference`` with ``where`` stated.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.time_reference [frames ...]
"""
from __future__ import annotations

import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_model() -> str:
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def main() -> int:
    import torch
    from oracle import ref_harness as rh
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    frames = [int(a) for a in sys.argv[1:]] or [200, 401]
    sd = make_state_dict(0, variant='peaky')
    model = rh.build_reference_model(sd, mode='RAW', bits=10)
    ncpu = os.cpu_count() or 1
    runs = []
    for T in frames:
        mels = make_mels(1234, 1, T)
        for threads in (ncpu, 1):
            out = rh.reference_generate(model, mels, seed=42, num_threads=threads)
            L = T * 275
            runs.append(dict(frames=T, loop_steps=L, threads=threads, seconds=round(out['seconds'], 3),
                             ksamples_per_s=round(L / out['seconds'] / 1000.0, 4)))
            print(runs[-1], flush=True)
    rec = dict(what='unmodified reference WaveRNN.generate() (PyTorch CPU, fatchord_version.py:169-264), unbatched, '
                    'RAW 10-bit, seeded synthetic weights (peaky) and mel, torch.manual_seed(42), stdout captured, wav write stubbed',
               where='build container (the GPU box has no /root/reference)', nproc=ncpu, cpu=cpu_model(),
               torch=torch.__version__, date=time.strftime('%Y-%m-%d'), runs=runs)
    path = os.path.join(ROOT, 'profiles', 'cpu_reference_container.json')
    with open(path, 'w') as f:
        json.dump(rec, f, indent=1)
    print('wrote', path)
    return 0


if __name__ == '__main__':
    sys.exit(main())
