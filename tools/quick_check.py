"""Developer probe (GPU box): team vs simple kernel agreement + timings.  Not part of the product."""
import sys, time
import numpy as np
import torch

sys.path.insert(0, '.')
from tacotronv2_wavernn_chinese_amd import _cabi
from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN


def model(mode='RAW', variant='peaky'):
    bits = 10 if mode == 'RAW' else 9
    sd = make_state_dict(0, mode=mode, variant=variant, bits=bits)
    dims = dict(DEFAULT_DIMS); dims['bits'] = bits
    m = WaveRNN(**dims, mode=mode)
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.to('cuda:0')


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 41
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    modes = ('RAW',) if len(sys.argv) > 3 and sys.argv[3] == 'raw' else ('RAW', 'MOL')
    kernels = (('team2', _cabi.KERNEL_TEAM2),) if len(sys.argv) > 4 and sys.argv[4] == 'team2' else (('batch', _cabi.KERNEL_BATCH), ('team2', _cabi.KERNEL_TEAM2)) if len(sys.argv) > 4 and sys.argv[4] == 'batch' else (('simple', _cabi.KERNEL_SIMPLE), ('batch', _cabi.KERNEL_BATCH), ('team2', _cabi.KERNEL_TEAM2))
    for mode in modes:
        m = model(mode)
        mels = make_mels(3, B, T)
        out = {}
        for name, k in kernels:
            try:
                t0 = time.time()
                r = m.generate_raw(mels, False, 11000, 550, noise_mode=_cabi.NOISE_PHILOX, seed=77, kernel=k)
                dt = time.time() - t0
                tm = m.last_timing
                out[name] = (r['labels'].cpu().numpy(), r['samples'].cpu().numpy())
                steps = tm['steps'] * tm['rows']
                print(f'{mode} {name:6s} B={B} T={T}: loop {tm["loop_ms"]:.2f} ms prologue {tm["prologue_ms"]:.3f} ms '
                      f'-> {steps / tm["loop_ms"]:.1f} ksamples/s, {tm["loop_ms"] * 1e3 / tm["steps"]:.2f} us/step (wall {dt:.2f}s)')
            except Exception as e:  # noqa
                print(f'{mode} {name}: FAILED {e!r}')
        names = list(out)
        for other in names[1:]:
            a, b = out[names[0]], out[other]
            print(f'  {names[0]} vs {other}:', end=' ')
            if mode == 'RAW':
                mism = np.argwhere(a[0] != b[0])
                print(f'  labels equal: {mism.size == 0}; first mismatch {mism[0] if mism.size else None}; n={len(mism)}')
            else:
                print(f'  max |sample diff| {np.abs(a[1] - b[1]).max():.3e}, mix idx equal {(a[0] == b[0]).all()}')


if __name__ == '__main__':
    main()
