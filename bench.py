"""bench.py -- the driver's benchmark contract for the WaveRNN mel->wav hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: the
prologue + per-sample loop for ONE ~5 s utterance per GPU (BASELINE.json
configs[1]: 1xMI355X, batch=1, 80-dim mel, T=401 frames -> 110 275 loop steps,
RAW 10-bit, hop 275), inputs resident in HBM when the timed region starts.
metric = audio ksamples/s (all loop steps counted, like the reference's own
"Gen Rate" meter, fatchord_version.py:267-271), whole job over all N GPUs
(one independent utterance per GPU per step: weak scaling, no data-path
collective -- utterances are independent).

Extra objects on the JSON line:
  roofline     -- memory-bound roofline of the batch-1 loop kernel: ALGORITHMIC bytes
                  (17 371 136 B of fp32 loop parameters + 836 B conditioning/sample per
                  step) x steps per launch / the loop kernel's average launch duration
                  (HIP events recorded by the library on the launch stream) vs 8 TB/s.
  cpu_baseline -- the CPU restatement (oracle/, "port") timed on this box's host cores on
                  a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_FRAMES = 401                      # ~5 s clip: wave_len = 400 * 275 = 110 000 samples = 4.989 s
HOP = 275
SAMPLE_RATE = 22050
BYTES_PER_SAMPLE_B1 = 17_371_136 + 836   # W + (C + O) at B = 1
FLOP_PER_SAMPLE = 8_668_160
HBM_PEAK = 8.0e12                   # MI355X_MICROARCH.md: 8 TB/s spec
# HBM bytes of the loop kernel over ONE utterance of the T=401 workload from the PMC passes in
# profiles/r01_rocprofv3_bench_team*.txt (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE)
TRAFFIC_BYTES_PER_UTTERANCE = {2: int((20082.2 * 2 + 1309.6) * 1024),           # team : weights once + per-frame records (1 launch)
                               3: int((542750.375 * 2 + 24372.96875) * 1024)}   # team2: 14 launches: weights 14x + the 8 KB/step stream + state


def cpu_baseline(frames: int = 61, max_threads: int = 16) -> dict:
    """Oracle (C port of the reference algorithm) on the host cores, bounded sample."""
    from oracle import oracle as orc
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    om = orc.OracleModel(sd, fast=True)
    threads = max(1, min(max_threads, os.cpu_count() or 1, om.num_threads()))
    mels = make_mels(1234, 1, frames)
    t0 = time.perf_counter()
    cm, ca = om.conditioning(mels)
    L = cm.shape[1]
    rng = np.random.Generator(np.random.PCG64(5))
    q = rng.standard_exponential((L, 1, 1024)).astype(np.float32)
    t1 = time.perf_counter()
    om.loop(cm, ca, orc.NOISE_EXPO, q, num_threads=threads)
    t2 = time.perf_counter()
    return dict(value=round(L / (t2 - t1) / 1000.0, 3), unit='ksamples/s', cores=threads, kind='port',
                sample=f'C restatement of generate() (oracle/wavernn_oracle.c, OpenMP+AVX2), B=1, mel 80x{frames} '
                       f'({L} loop steps, {t2 - t1:.1f} s loop + {t1 - t0:.1f} s prologue/noise), RAW 10-bit, injected Exp(1) noise')


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=T_FRAMES)
    ap.add_argument('--kernel', default='auto', choices=['auto', 'team2', 'team', 'simple'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})',
              file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    # synthetic weights of the reference architecture + synthetic mel (no checkpoint ships: .MISSING_LARGE_BLOBS)
    sd = make_state_dict(0, variant='peaky')
    model = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    model.verbose = False
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model.to(dev)
    model.kernel = {'auto': _cabi.KERNEL_AUTO, 'team2': _cabi.KERNEL_TEAM2, 'team': _cabi.KERNEL_TEAM, 'simple': _cabi.KERNEL_SIMPLE}[args.kernel]
    T = args.frames
    mels = torch.from_numpy(make_mels(1000 + rank, 1, T)).to(dev)   # resident in HBM before timing
    nat = model.native()
    rows, L = nat.plan(1, T, False, 11000, 550)
    samples = torch.empty((rows, L), dtype=torch.float32, device=dev)
    labels = torch.empty((rows, L), dtype=torch.int32, device=dev)
    wave_len = (T - 1) * HOP
    wave = torch.empty((wave_len,), dtype=torch.float64, device=dev)   # what generate() returns (:264)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def one_step(i: int):
        nat.generate(mels.data_ptr(), 1, T, False, 11000, 550, labels_ptr=labels.data_ptr(),
                     samples_ptr=samples.data_ptr(), stream=stream, noise_mode=_cabi.NOISE_PHILOX,
                     seed=0xC0FFEE + 7919 * i + rank, kernel=model.kernel)
        # float64 tail of generate() (mu-law decode, trim, fade-out) on the device: the step ends with the waveform
        nat.epilogue(samples.data_ptr(), labels.data_ptr(), rows, L, False, 11000, 550, True, wave_len,
                     wave.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        one_step(-1 - i)
    barrier()
    loop_ms, pro_ms = [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
        tm = nat.last_timing()          # waits for this step's kernels (HIP events on the launch stream)
        loop_ms.append(tm['loop_ms'])
        pro_ms.append(tm['prologue_ms'])
    barrier()
    dt = time.perf_counter() - t0
    kernel_ran = tm['kernel']
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_samples = world * args.steps * rows * L
        value = total_samples / dt / 1000.0
        audio_s = (T - 1) * HOP / SAMPLE_RATE
        k_ms = float(np.mean(loop_ms))
        launches = max(1, int(tm.get('launches', 1)))    # segments one utterance is generated in
        achieved = (BYTES_PER_SAMPLE_B1 * rows * L / launches) / (k_ms * 1e-3 / launches)
        traffic = TRAFFIC_BYTES_PER_UTTERANCE.get(kernel_ran) if T == T_FRAMES else None
        out = {
            'metric': 'audio ksamples/sec (22.05 kHz, 10-bit RAW WaveRNN, batch=1 per GPU)',
            'value': round(value, 3), 'unit': 'ksamples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[1]: 1 utterance per GPU per step, B=1, mel 80x{T} '
                                   f'({L} loop steps = {audio_s:.3f} s audio), RAW 10-bit, hop 275, prologue + loop + float64 epilogue on the device, '
                                   'Philox sampling noise, seeded synthetic weights (fc3 x128)',
                       'kernel': {1: 'simple', 2: 'team', 3: 'team2'}.get(kernel_ran, str(kernel_ran)),
                       'real_time_factor': round((dt / args.steps) / audio_s, 4),
                       'times_real_time': round(audio_s / (dt / args.steps), 2),
                       'prologue_ms': round(float(np.mean(pro_ms)), 3), 'loop_kernel_ms': round(k_ms, 3),
                       'loop_launches_per_utterance': launches, 'loop_launch_avg_ms': round(k_ms / launches, 3),
                       'parallelism': f'utterance-parallel x{world} (no data-path collective)'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved / 1e9, 2), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK, 4), 'traffic': None if traffic is None else traffic // launches,
                         'note': 'achieved = algorithmic bytes (17 371 972 B/sample at B=1) x steps per launch / average loop-kernel '
                                 'launch duration (an utterance is generated in loop_launches_per_utterance launches; HIP events '
                                 'around the whole sequence / launches); traffic = measured HBM bytes per launch (PMC passes, '
                                 'profiles/): the weights are register/LDS resident, so the kernel is latency/issue-bound, not '
                                 'HBM-bound'},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline()
            except Exception as e:  # the baseline must never sink the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'ksamples/s', 'cores': 0, 'kind': 'port',
                                       'sample': f'failed: {e!r}'}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
