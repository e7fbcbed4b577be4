"""bench.py -- the driver's benchmark contract for the WaveRNN mel->wav hot path.

    python bench.py --gpus N --steps K --warmup W [--config {1,2,4,3}]

`python bench.py --gpus N` launches its own N ranks (re-exec under torch.distributed.run, one rank per GPU, RCCL);
started under torch.distributed.run it uses the ranks it is given (`--gpus` then defaults to WORLD_SIZE).

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
  --config 1 (default; BASELINE.json configs[1], the config the metric is quoted on): ONE ~5 s utterance per GPU,
             B=1, mel 80x401 -> 110 275 loop steps, RAW 10-bit, hop 275: prologue + per-sample loop (latency kernel,
             one XCD team) + float64 epilogue on the device;
  --config 2 (configs[2]): 64 utterances of that shape per GPU in one call (batch kernel: 8 rows per XCD team in lock-step
             on the matrix cores);
  --config 4 (configs[4]): MOL 9-bit, 32 utterances per GPU (batch kernel, 4 rows per team);
  --config 3 (configs[3]): throughput mode, 64 utterances per GPU (512 over 8 GPUs): rank 0 owns all clips, scatters the
             mels over RCCL, every rank generates its 64, the waveforms are gathered back on rank 0 -- the only collectives
             of the path (utterances are independent: SURVEY.md 8e); scatter + gather are inside the timed region.  The
             process group is ALWAYS initialised for this config, also at N=1 (a one-rank RCCL communicator): the scatter /
             gather calls are the same code at every N.
metric = audio ksamples/s (all loop steps of all rows counted, like the reference's own "Gen Rate" meter,
fatchord_version.py:267-271), whole job over all N GPUs; weak scaling (the per-GPU batch is fixed).

The headline fields are the selected config's.  The default run (config 1, N=1) additionally times configs[2] and [4] for a
few steps and attaches them as `extra_configs` {"2": {...}, "4": {...}} to the same JSON line (~10 s), plus "fold_auto": ONE 5 s utterance
in the reference's fold mode with target='auto' (the latency-optimal fold count: single-utterance latency), "fold_hp_defaults" (the
reference's own target 11000 / overlap 550) and "fold_per_xcd" (one fold per XCD team: rounds 4-5's 'auto'), "dm": the secondary
dual-softmax model (deepmind_version.py:75-165; 50 000 samples) and "train_step": the training step of the loop
layers (`wrnn_train_step`) at the reference's training shape (~1 s each).  A run on N > 1
GPUs (config 1) attaches `extra_configs` {"3": {...}}: BASELINE configs[3] (64 clips per GPU, scatter / gather over RCCL) at that N.

Extra objects on the JSON line:
  roofline      -- ALGORITHMIC bytes (weights once per step for the whole batch + 836 B conditioning/sample, SURVEY.md
                   8d) or FLOPs (8 668 160 per sample RAW) x steps per launch / the loop kernel's average launch duration
                   (HIP events recorded by the library on the launch stream), against the bound SURVEY 8d names for the
                   batch size: HBM for B=1 and MOL B=32 (`peak` = the 8 TB/s of the data sheet, `peak_measured` = a
                   read-only stream over 1 GiB timed in this run, `peak_measured_copy` = a device-to-device copy, read + write
                   bytes), the fp32 matrix/vector peak 157.3 TFLOP/s
                   for B=64.  `frac_of_f32_peak` = the same rate against the fp32 pipe for every config; `traffic` = measured HBM bytes per launch and
                   `mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1 024 SIMDs) from the PMC passes under profiles/
                   (static numbers of that profile session: `traffic_source` names file + commit; null when the kernel sources changed since).  For B=1 the weights are
                   register/LDS resident and HBM is idle: the bound that actually binds is the exchange latency,
                   `latency_model` = {exchanges per step, all-gather round time of the 32 workgroups of one XCD measured by
                   bench_micro/handoff, floor_us} and `frac_of_latency_floor` = floor / measured us per step.
  cpu_baseline  -- the reference's generate() loop issued op for op on PyTorch-CPU on THIS box's host cores (oracle/torch_cpu_loop.py,
                   kind "reference-ops (torch CPU, this box)": ALL 55 000 steps of BASELINE configs[0]'s clip at <= 16 threads, samples at torch's
                   untuned default thread count and at one thread), rank 0 at N=1 only;  cpu_port -- the C restatement (oracle/wavernn_oracle.c, OpenMP + AVX2) on the same clip;
  cpu_reference -- the UNMODIFIED reference generate() (PyTorch CPU) timed by oracle/time_reference.py where
                   /root/reference exists (the build container; `where` says so) -- the GPU box has no reference tree.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_FRAMES = 401                      # ~5 s clip: wave_len = 400 * 275 = 110 000 samples = 4.989 s
HOP = 275
SAMPLE_RATE = 22050
W_BYTES = {'RAW': 17_371_136, 'MOL': 15_331_448}       # fp32 loop parameters
COND_BYTES = 836                                        # conditioning row + sample per loop step
FLOP_PER_SAMPLE = {'RAW': 8_668_160, 'MOL': 7_650_304}
HBM_PEAK = 8.0e12                   # MI355X_MICROARCH.md: 8 TB/s spec
F32_PEAK = 157.3e12                 # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
SCALE_LEG_TIMEOUT = float(os.environ.get('WRNN_BENCH_SCALE_LEG_TIMEOUT', '180'))   # seconds the configs[3] leg of an N > 1 run may take before the watchdog prints the headline without it
CONFIGS = {1: dict(mode='RAW', bits=10, batch=1, name='configs[1]'),
           2: dict(mode='RAW', bits=10, batch=64, name='configs[2]'),
           4: dict(mode='MOL', bits=9, batch=32, name='configs[4]'),
           3: dict(mode='RAW', bits=10, batch=64, name='configs[3]')}
# Counter numbers of the loop kernels (HBM bytes and MFMA-busy cycles per launch) come from the rocprofv3 --pmc passes of tools/profile_round.sh,
# kept as profiles/<round>_pmc.json by tools/pmc_summary.py --json: STATIC numbers of that profile session, not of this run.  The file records the
# commit it was taken at and a hash of the kernel sources (csrc/*.hip, *.h, include/wavernn_amd.h); when the tree's sources hash differently -- a
# kernel was edited after the last profile -- the counter fields are reported as null instead of silently describing another kernel.
PMC_FILE = next((f for f in (os.path.join('profiles', 'r06_pmc.json'), os.path.join('profiles', 'r05_pmc.json'))
                 if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), f))), os.path.join('profiles', 'r06_pmc.json'))   # the newest counter file in the tree
ENGINE_HZ, N_SIMD = 2.4e9, 1024      # MI355X_MICROARCH.md: 2.4 GHz peak engine clock, 256 CUs x 4 SIMDs


def csrc_sha() -> str:
    """sha256 over the kernel sources in the tree (names + contents, sorted): what the counter file must have been taken on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'tacotronv2_wavernn_chinese_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h')) + [os.path.join(ROOT, 'include', 'wavernn_amd.h')]):
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc() -> dict:
    """{'ok': bool, 'why': str, 'source': str, 'configs': {cfg_id: {...}}} -- ok only if the file exists and was taken on these sources."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return dict(ok=False, why=f'{PMC_FILE} not present', source=None, configs={})
    src = (f'{PMC_FILE} (static: rocprofv3 --pmc passes of tools/profile_round.sh taken at commit {rec.get("commit")}, kernel sources sha256 {rec.get("csrc_sha")}'
           ' -- not this run)')
    now = csrc_sha()
    if rec.get('csrc_sha') != now:
        return dict(ok=False, why=f'{PMC_FILE} was taken on kernel sources {rec.get("csrc_sha")} (commit {rec.get("commit")}), the tree has {now}: counters withheld',
                    source=src, configs={})
    return dict(ok=True, why='', source=src, configs={int(k): v for k, v in rec.get('configs', {}).items()})


# What bounds the B=1 latency kernel: 4 dependent all-gathers among the 32 workgroups of one XCD per step.
# bench_micro/handoff.hip measures one such round (512 granules published, polled with sc1 loads, written to LDS, 2
# barriers, NO compute between rounds): profiles/r03_handoff_microbench.txt.
DM_EXCHANGES = 6   # all-gather rounds per sample of dm_team_kernel
LATENCY_MODEL = dict(exchanges_per_step=4, barriers_per_step=5, allgather_round_us=0.746, raw_hop_us=0.27,
                     source='profiles/r03_handoff_microbench.txt (gather st=plain ld=sc1 team=32: 0.746-0.768 us per round; '
                            'raw one-way hop profiles/r01_handoff_microbench.txt)')


def cpu_reference_ops(frames: int = 200) -> dict:
    """The reference's generate() loop issued op for op on PyTorch-CPU ON THIS BOX (oracle/torch_cpu_loop.py: nn.GRUCell /
    nn.Linear / softmax / Categorical.sample() as fatchord_version.py:194-237; pinned to the reference's own labels by
    tests/test_oracle_golden.py) on BASELINE configs[0]'s clip: ALL of its 55 000 loop steps (wavernn_gen.py:126 runs the whole clip) at the
    thread count that suits this box (<= 16), plus bounded samples at torch's untuned default (one thread per core) and at one thread."""
    from oracle import torch_cpu_loop as tl
    from tacotronv2_wavernn_chinese_amd.synth import make_mels, make_state_dict
    sd = make_state_dict(0, variant='peaky')
    mels = make_mels(1234, 1, frames)
    ncpu = os.cpu_count() or 1
    L = frames * HOP
    tuned = min(ncpu, 16)
    tl.run(sd, mels, 32, tuned, max_seconds=5.0)   # first-call overheads (thread pool, MKL plans)
    full = tl.run(sd, mels, L, tuned, max_seconds=90.0)                  # the whole clip (~20 s on the GPU box's host)
    # torch's default is one thread per core; on a many-core host that is the WORST setting for 512-wide matrix-vector products (every op is
    # an OpenMP barrier): what `python wavernn_gen.py` does untuned on this box -- reported as it is, next to the tuned figure
    dflt = tl.run(sd, mels, 3000, ncpu, max_seconds=8.0) if ncpu != tuned else full
    one = tl.run(sd, mels, 3000, 1, max_seconds=6.0)
    rate = lambda r: round(r['steps'] / max(r['loop_seconds'], 1e-9) / 1000.0, 4)
    whole = full['steps'] == L
    return dict(value=rate(full), unit='ksamples/s', cores=tuned, host_cores=ncpu,
                by_threads={str(tuned): rate(full), str(ncpu): rate(dflt), '1': rate(one)},
                untuned_default_threads=dict(threads=ncpu, value=rate(dflt), steps=dflt['steps']),
                kind='reference-ops (torch CPU, this box)', one_thread=rate(one),
                sample=(f'ALL {L} loop steps' if whole else f'the first {full["steps"]} of the {L} loop steps (90 s cap)') +
                       f' of BASELINE configs[0] (mel 80x{frames}, RAW 10-bit, B=1) with the reference\'s own op sequence on torch-CPU: {full["loop_seconds"]:.1f} s on {tuned} '
                       f'threads of {ncpu} cores; torch\'s default of {ncpu} threads: {dflt["steps"]} steps in {dflt["loop_seconds"]:.1f} s; 1 thread: {one["steps"]} steps in '
                       f'{one["loop_seconds"]:.1f} s; oracle/torch_cpu_loop.py, reproduces the reference\'s labels bit for bit')


def cpu_baseline(frames: int = 200, max_threads: int = 16) -> dict:
    """Oracle (C port of the reference algorithm) on the host cores: BASELINE configs[0]'s clip (one 80x200 mel)."""
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
                sample=f'BASELINE configs[0]: C restatement of generate() (oracle/wavernn_oracle.c, OpenMP+AVX2), B=1, mel 80x{frames} '
                       f'({L} loop steps, {t2 - t1:.1f} s loop + {t1 - t0:.1f} s prologue/noise), RAW 10-bit, injected Exp(1) noise')


def cpu_reference() -> dict | None:
    """The unmodified reference's own timing (oracle/time_reference.py, run where /root/reference exists)."""
    path = os.path.join(ROOT, 'profiles', 'cpu_reference_container.json')
    try:
        with open(path) as f:
            rec = json.load(f)
    except OSError:
        return None
    runs = {(r['frames'], r['threads']): r for r in rec['runs']}
    allc, one = runs.get((200, rec['nproc'])), runs.get((200, 1))
    if not allc:
        return None
    out = dict(value=allc['ksamples_per_s'], unit='ksamples/s', cores=rec['nproc'], kind='reference',
               one_thread=one['ksamples_per_s'] if one else None, where=rec['where'], cpu=rec['cpu'], measured=rec['date'],
               sample=f'BASELINE configs[0]: {rec["what"]}; mel 80x200 ({allc["loop_steps"]} loop steps) in {allc["seconds"]} s on '
                      f'{rec["nproc"]} threads' + (f', {one["seconds"]} s on 1 thread' if one else ''))
    big = runs.get((401, rec['nproc']))
    if big:
        out['config1_clip'] = dict(frames=401, ksamples_per_s=big['ksamples_per_s'], seconds=big['seconds'])
    return out


def measure_hbm_peaks(dev, nbytes: int = 1 << 30, reps: int = 8) -> dict | None:
    """Device-memory bandwidth of THIS box ("replace nominal with a measured device-copy peak on the box"), B/s:
    `read` = a read-only float4 stream, `copy` = device-to-device copy (read + write bytes).  Measured by the hand-written
    kernels of bench_micro/devcopy.hip (built by __graft_entry__.build(); HIP events around 10 passes over 2 GiB; runs on
    device 0 = rank 0's GPU); if that binary is missing, a torch device copy timed with HIP events stands in for both."""
    import re
    exe = os.path.join(ROOT, 'bench_micro', 'devcopy')
    try:
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
        copies = [float(x) for x in re.findall(r'copy\s+grid\s+\d+:\s+([0-9.]+) GB/s', txt)]
        reads = [float(x) for x in re.findall(r'read\s+grid\s+\d+:\s+([0-9.]+) GB/s', txt)]
        if copies and reads:
            return {'copy': max(copies) * 1e9, 'read': max(reads) * 1e9, 'how': 'bench_micro/devcopy (hand-written float4 stream kernels, 2 GiB)'}
    except (OSError, subprocess.SubprocessError):
        pass
    import torch
    try:
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(dev)
        bw = 2.0 * nbytes / (e0.elapsed_time(e1) / reps * 1e-3)
        del src, dst
        return {'copy': bw, 'read': bw, 'how': 'torch device copy (read + write), bench_micro/devcopy not built'}
    except Exception:   # the peak is an annotation: never sink the bench line for it
        return None


def train_step_leg(dev, B: int = 32, frames: int = 5, iters: int = 6) -> dict:
    """SURVEY.md 8f N4, driver-visible: `wrnn_train_step` (forward + the training script's loss + backward of the loop layers) at the
    reference's own training shape (voc_batch_size 32 x voc_seq_len 5 hops, wavernn_hparams.py:44,51) on synthetic conditioning, timed with
    events on the launch stream.  The upsample network and the optimiser (torch ops around it in a training iteration) are not in it."""
    import numpy as np
    import torch
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    torch.manual_seed(0)
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.verbose = False
    m.to(dev).train()
    L = frames * DEFAULT_DIMS['hop_length']
    rng = np.random.Generator(np.random.PCG64(1))
    x = torch.from_numpy(rng.uniform(-1, 1, (B, L)).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.integers(0, 2 ** DEFAULT_DIMS['bits'], (B, L)).astype(np.int32)).to(dev)
    mu = torch.from_numpy(rng.random((B, L, DEFAULT_DIMS['feat_dims']), dtype=np.float32)).to(dev)
    au = torch.from_numpy(rng.standard_normal((B, L, DEFAULT_DIMS['res_out_dims'])).astype(np.float32)).to(dev)
    ps = [p.detach().contiguous() for p in m._loop_params()]
    gs = [torch.zeros_like(p) for p in ps]
    dm, da = torch.zeros_like(mu), torch.zeros_like(au)
    loss = torch.zeros((), device=dev)
    nat = m._native_handle()
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream

        def call():
            nat.train_step([p.data_ptr() for p in ps], [g.data_ptr() for g in gs], x.data_ptr(), mu.data_ptr(), au.data_ptr(), y.data_ptr(), B, L,
                           loss.data_ptr(), 0, dm.data_ptr(), da.data_ptr(), st)
        for _ in range(2):
            call()
        nat.sync_status(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            call()
        e1.record()
        nat.sync_status(st)
        ms = e0.elapsed_time(e1) / iters
    value = float(loss)
    del m, nat, ps, gs, dm, da, mu, au
    torch.cuda.empty_cache()
    return {'metric': 'training step of the loop layers (forward + loss + backward, wrnn_train_step), audio ksamples/sec', 'value': round(B * L / ms, 1),
            'unit': 'ksamples/s', 'steps': iters, 'warmup': 2, 'ms_per_step': round(ms, 3), 'dtype': 'f32',
            'config': {'workload': f'B={B} x L={L} teacher-forced steps (voc_batch_size x voc_seq_len of the reference), RAW 10-bit, synthetic '
                                   'conditioning and targets, seeded random weights', 'loss': round(value, 6)}}


def fold_leg(dev, target='auto', overlap: int = 550, frames: int = T_FRAMES, reps: int = 7, warm: int = 3) -> dict:
    """The reference's own fast mode ("batched ... very fast (realtime+)", wavernn_hparams.py:55-57, fatchord_version.py:293-405) as a
    single-utterance LATENCY figure: one 5 s clip through generate(batched=True, target, overlap), crossfaded and unfolded on the device;
    value = wave_len / wall time of the whole generate() call (upload, prologue, loop, epilogue, download, wav file): the MEDIAN of `reps` calls
    (a latency figure; every call's time, the mean and the minimum are reported next to it).  `warm` untimed calls first: the first calls of a fresh
    model carry one-off stalls of 10-60 ms inside the enqueue / wait (not GC, not the wav file, loop-kernel time unchanged: the HIP runtime growing its
    pools, profiles/r06_fold_spikes.txt) -- BENCH_r05's 65 ms "mean of 3" was one such call among 48 ms ones.
    target='auto': the fold count with the lowest predicted loop time (vocoder.fold_plan: 64 folds on the batch kernel for this clip);
    target=11000: the reference's hp defaults (10 folds); target='per_xcd': one fold per XCD team on the latency kernel (rounds 4-5's 'auto')."""
    import tempfile
    import torch
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    sd = make_state_dict(0, variant='peaky')
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.verbose = False
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m.to(dev)
    mels = make_mels(1000, 1, frames)
    wave_len = (frames - 1) * HOP
    per_call, loops = [], []
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, 'o.wav')
        for i in range(warm):
            m.generate(mels, path, True, target, overlap, True, epilogue='device', seed=100 + i)
        torch.cuda.synchronize(dev)
        for i in range(reps):
            t0 = time.perf_counter()
            wav = m.generate(mels, path, True, target, overlap, True, epilogue='device', seed=2 + i)
            torch.cuda.synchronize(dev)
            per_call.append((time.perf_counter() - t0) * 1e3)
            loops.append(m.last_timing['loop_ms'])
    dt = float(np.median(per_call)) * 1e-3
    tm = m.last_timing
    assert wav.shape == (wave_len,)
    what = {'auto': 'target="auto" (the cost model\'s fold count)', 'per_xcd': 'target="per_xcd" (one fold per XCD team)'}.get(target, f'target={target} (the reference\'s hp defaults)')
    out = {'metric': f'single-utterance latency mode: audio ksamples/sec of ONE clip in fold mode, {what}',
           'value': round(wave_len / dt / 1000.0, 1), 'unit': 'ksamples/s', 'steps': reps, 'warmup': warm, 'ms_per_step': round(dt * 1e3, 3), 'dtype': 'f32',
           'config': {'workload': f'one utterance, mel 80x{frames} ({wave_len / SAMPLE_RATE:.3f} s of audio), generate(batched=True, target={target!r}, overlap={overlap}), '
                                  f'{tm["rows"]} folds x {tm["steps"]} loop steps, RAW 10-bit, device epilogue (crossfade + unfold), wav written',
                      'times_real_time': round(wave_len / SAMPLE_RATE / dt, 1), 'loop_kernel_ms': round(float(np.mean(loops)), 3),
                      'loop_share_of_call': round(float(np.mean(loops)) / (dt * 1e3), 4), 'ms_per_call': [round(x, 3) for x in per_call],
                      'ms_median': round(dt * 1e3, 3), 'ms_mean': round(float(np.mean(per_call)), 3), 'ms_min': round(min(per_call), 3), 'prologue_ms': round(tm['prologue_ms'], 3), 'kernel': _kernel_name(tm['kernel'])}}
    del m
    torch.cuda.empty_cache()
    return out


def dm_leg(dev, n: int = 50_000) -> dict:
    """The secondary dual-softmax model (deepmind_version.py:75-165) on the driver-visible line: wrnn_dm_generate,
    hidden 896, `n` samples on the team kernel (one XCD), wall time of the call (launch + loop + sync)."""
    import torch
    from tacotronv2_wavernn_chinese_amd.deepmind import WaveRNN as DM
    from tacotronv2_wavernn_chinese_amd.synth import make_dm_state_dict
    m = DM()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in make_dm_state_dict(0).items()})
    m.to(dev)
    m.generate(2000, seed=1, kernel=2)
    t0 = time.perf_counter()
    out, c, f = m.generate(n, seed=2, kernel=2)
    dt = time.perf_counter() - t0
    us = dt / n * 1e6
    ex = DM_EXCHANGES
    floor = ex * LATENCY_MODEL['allgather_round_us']
    res = {'metric': 'secondary dual-softmax WaveRNN (16-bit, coarse/fine), audio ksamples/sec, batch=1', 'value': round(n / dt / 1000.0, 1), 'unit': 'ksamples/s',
           'steps': 1, 'warmup': 1, 'ms_per_step': round(dt * 1e3, 3), 'dtype': 'f32',
           'config': {'workload': f'deepmind_version.WaveRNN(hidden 896, quantisation 256).generate({n}), seeded synthetic weights, Philox noise, team kernel (32 CUs of one XCD)',
                      'us_per_sample': round(us, 4), 'distinct_coarse': int(len(np.unique(c))), 'distinct_fine': int(len(np.unique(f)))},
           'roofline': {'bound': 'latency', 'latency_model': {'exchanges_per_sample': ex, 'allgather_round_us': LATENCY_MODEL['allgather_round_us'], 'floor_us': round(floor, 3),
                                                             'source': LATENCY_MODEL['source']},
                        'frac_of_latency_floor': round(floor / us, 4)}}
    del m
    torch.cuda.empty_cache()
    return res


def _kernel_name(k: int) -> str:
    from tacotronv2_wavernn_chinese_amd import _cabi
    return _cabi.KERNEL_NAMES.get(k, str(k))


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no ranks around it: become the launcher."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def run_config(cfg_id: int, *, world: int, rank: int, dev, dry: bool, steps: int, warmup: int, frames: int, batch: int,
               kernel_name: str, copy_peak: dict | None, dump: str | None = None, phase_profile: bool = False) -> dict | None:
    """Time `steps` passes of config `cfg_id` (after `warmup` untimed ones) and return the fields of its JSON line (rank 0;
    None elsewhere)."""
    import torch
    import torch.distributed as dist
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN

    cfg = CONFIGS[cfg_id]
    mode, B = cfg['mode'], (batch or cfg['batch'])
    use_pg = dist.is_initialized()
    # synthetic weights of the reference architecture + synthetic mel (no checkpoint ships: .MISSING_LARGE_BLOBS)
    sd = make_state_dict(0, mode=mode, variant='peaky' if mode == 'RAW' else 'default', bits=cfg['bits'])
    dims = dict(DEFAULT_DIMS)
    dims['bits'] = cfg['bits']
    model = WaveRNN(**dims, mode=mode)
    model.verbose = False
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    if not dry:
        model.to(dev)
    kernel = _cabi.KERNEL_IDS[kernel_name]
    T = frames
    scatter = cfg_id == 3
    if scatter and rank == 0:
        all_mels = torch.from_numpy(make_mels(1000, world * B, T)).to(dev)   # rank 0 owns the whole job, resident in HBM
        mels = torch.empty((B, 80, T), dtype=torch.float32, device=dev)
    elif scatter:
        all_mels, mels = None, torch.empty((B, 80, T), dtype=torch.float32, device=dev)
    else:
        all_mels = None
        mels = torch.from_numpy(make_mels(1000 + rank, B, T)).to(dev)       # resident in HBM before timing
    nat = None if dry else model.native()
    if phase_profile and nat is not None:
        nat.phase_profile(True)
    rows, L = (B, T * HOP) if dry else nat.plan(B, T, False, 11000, 550)
    samples = torch.empty((rows, L), dtype=torch.float32, device=dev)
    labels = torch.empty((rows, L), dtype=torch.int32, device=dev)
    wave_len = (T - 1) * HOP
    wave = torch.empty((rows, wave_len), dtype=torch.float64, device=dev)   # what generate() returns (:264), per utterance
    gathered = [torch.empty((rows, wave_len), dtype=torch.float64, device=dev) for _ in range(world)] if (scatter and rank == 0) else None
    stream = 0 if dry else torch.cuda.current_stream(dev).cuda_stream

    def seed_of(i: int) -> int:
        return 0xC0FFEE + 7919 * i + rank

    def one_step(i: int):
        if scatter:   # same calls at every N (a one-rank communicator at N=1)
            dist.scatter(mels, list(all_mels.view(world, B, 80, T).unbind(0)) if rank == 0 else None, src=0)
        if dry:   # stand-in for the device calls: every utterance's "waveform" is a function of its own mel only
            wave.copy_(mels.to(torch.float64).sum(dim=(1, 2))[:, None].expand(rows, wave_len))
        else:
            nat.generate(mels.data_ptr(), B, T, False, 11000, 550, labels_ptr=labels.data_ptr(),
                         samples_ptr=samples.data_ptr(), stream=stream, noise_mode=_cabi.NOISE_PHILOX,
                         seed=seed_of(i), kernel=kernel)
            # float64 tail of generate() (mu-law decode, trim, fade-out) on the device for every utterance of the batch, one launch
            nat.epilogue_rows(samples.data_ptr(), labels.data_ptr(), rows, L, mode == 'RAW', wave_len, 0, wave.data_ptr(), wave_len, stream)
        if scatter:
            dist.gather(wave, gathered, dst=0)

    def barrier():
        if not dry:
            torch.cuda.synchronize(dev)
        if use_pg:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    for i in range(warmup):
        one_step(-1 - i)
    barrier()
    loop_ms, pro_ms = [], []
    tm = None
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(i)
        tm = dict(kernel=0, loop_ms=1.0, prologue_ms=0.0, launches=1) if dry else nat.last_timing()   # waits for this step's kernels (HIP events on the launch stream)
        loop_ms.append(tm['loop_ms'])
        pro_ms.append(tm['prologue_ms'])
    barrier()
    dt = time.perf_counter() - t0
    kernel_ran = tm['kernel'] if tm else 0
    if use_pg and world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    phases = None
    if phase_profile and nat is not None:
        phases = nat.phase_cycles()
        nat.phase_profile(False)
    if rank != 0:
        return None

    if dump:   # what the last timed step produced, for tests/test_gpu_config3.py
        np.savez(dump, wave=torch.stack(gathered).cpu().numpy() if gathered is not None else wave.cpu().numpy()[None],
                 labels=labels.cpu().numpy(), samples=samples.cpu().numpy(),   # rank 0's own rows
                 seed=np.asarray([seed_of(steps - 1)], np.uint64), frames=T, batch=B, world=world, mel_seed=1000)
    total_samples = world * steps * rows * L
    value = total_samples / dt / 1000.0
    audio_s = (T - 1) * HOP / SAMPLE_RATE
    k_ms = float(np.mean(loop_ms)) if loop_ms else float('nan')
    launches = max(1, int(tm.get('launches', 1))) if tm else 1   # segments an utterance is generated in
    bytes_per_sample = W_BYTES[mode] / rows + COND_BYTES
    bw_bound = HBM_PEAK / bytes_per_sample
    fl_bound = F32_PEAK / FLOP_PER_SAMPLE[mode]
    rate = rows * L / (k_ms * 1e-3)                              # row-steps per second of the loop kernel alone
    us_per_step = k_ms * 1e3 / L
    if fl_bound < bw_bound:
        roof = {'bound': 'mfma', 'achieved': round(rate * FLOP_PER_SAMPLE[mode] / 1e12, 3), 'peak': F32_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': round(rate / fl_bound, 4)}
        roof_note = (f'achieved = {FLOP_PER_SAMPLE[mode]} FLOP/sample x {rows} rows x steps per launch / loop-kernel launch duration; '
                     f'bound = fp32 matrix/vector peak (B={rows}: {fl_bound / 1e6:.1f} Msamples/s; the HBM bound would be {bw_bound / 1e6:.1f})')
    else:
        roof = {'bound': 'hbm', 'achieved': round(rate * bytes_per_sample / 1e9, 2), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                'frac': round(rate / bw_bound, 4)}
        if copy_peak:   # {'copy': ..., 'read': ...} B/s measured in this run; the algorithm's traffic is reads, so frac uses `read`
            roof['peak_measured'] = round(copy_peak['read'] / 1e9, 1)
            roof['peak_measured_copy'] = round(copy_peak['copy'] / 1e9, 1)
            roof['frac_of_measured'] = round(rate * bytes_per_sample / copy_peak['read'], 4)
            roof['peak_measured_by'] = copy_peak['how']
        roof_note = (f'achieved = algorithmic bytes ({bytes_per_sample:.0f} B/sample at B={rows}: weights once per step for the batch + 836 B) '
                     f'x steps per launch / average loop-kernel launch duration (HIP events around the {launches} launch(es) of one call); '
                     'peak = data-sheet HBM bandwidth, peak_measured = read-only stream / peak_measured_copy = device copy (read + write) timed in this run.  NOTIONAL for '
                     'this kernel: the weights are register/LDS resident and never leave the chip (traffic: measured HBM bytes per launch)')
    # the physically relevant ceilings next to the SURVEY-binding one: the fp32 matrix / vector pipe for every config ...
    roof['frac_of_f32_peak'] = round(rate * FLOP_PER_SAMPLE[mode] / F32_PEAK, 4)
    # ... and what the counters of the last profile session say (static, withheld when the kernel sources changed since: load_pmc)
    pmc = load_pmc()
    pc = pmc['configs'].get(2 if cfg_id == 3 else cfg_id) if (pmc['ok'] and T == T_FRAMES and B == cfg['batch']) else None
    if pc is not None and f'loop_{_kernel_name(kernel_ran)}_kernel<' not in pc.get('kernel', ''):   # exact kernel, not a substring ('batch' is in 'batch_cs')
        pc = None
    roof['traffic'] = (pc['fetch_bytes_per_launch'] + pc['write_bytes_per_launch']) if pc else None
    # matrix-pipe occupancy: SQ_VALU_MFMA_BUSY_CYCLES of one launch / (this run's launch duration x 2.4 GHz x 1 024 SIMDs)
    roof['mfma_busy_frac'] = round(pc['mfma_busy_cycles_per_launch'] / ((k_ms / launches) * 1e-3 * ENGINE_HZ * N_SIMD), 4) if pc else None
    roof['traffic_source'] = pmc['source'] if pc else (pmc['why'] or None)
    if kernel_ran == _cabi.KERNEL_TEAM2 and rows <= 8:
        lm = dict(LATENCY_MODEL)
        lm['floor_us'] = round(lm['exchanges_per_step'] * lm['allgather_round_us'], 3)
        roof['latency_model'] = lm
        roof['frac_of_latency_floor'] = round(lm['floor_us'] / us_per_step, 4)
        roof_note += ('; what binds is latency: latency_model.floor_us = 4 dependent all-gather rounds per step with no compute at all, '
                      'frac_of_latency_floor = floor_us / measured us per step')
    roof['note'] = roof_note
    out = {
        'metric': f'audio ksamples/sec (22.05 kHz, {"10-bit RAW" if mode == "RAW" else "9-bit MOL"} WaveRNN, batch={B} per GPU)',
        'value': round(value, 3), 'unit': 'ksamples/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'BASELINE {cfg["name"]}: {B} utterance(s) per GPU per step, mel 80x{T} '
                               f'({L} loop steps = {audio_s:.3f} s audio each), {"RAW 10-bit" if mode == "RAW" else "MOL 9-bit"}, hop 275, '
                               'prologue + loop + float64 epilogue on the device, Philox sampling noise, seeded synthetic weights'
                               + (', mels scattered from / waveforms gathered on rank 0 over RCCL inside the timed region' if scatter else ''),
                   'kernel': _cabi.KERNEL_NAMES.get(kernel_ran, str(kernel_ran)),
                   'real_time_factor': round((dt / steps) / (audio_s * rows), 5),
                   'times_real_time': round(audio_s * rows / (dt / steps), 2),
                   'prologue_ms': round(float(np.mean(pro_ms)), 3), 'loop_kernel_ms': round(k_ms, 3),
                   'loop_launches_per_call': launches, 'loop_launch_avg_ms': round(k_ms / launches, 3),
                   'us_per_step': round(us_per_step, 4),
                   'parallelism': f'utterance-parallel x{world} (no data-path collective' + (' but the scatter/gather of clips)' if scatter else ')')},
        'roofline': roof,
    }
    if phases is not None:
        out['phase_cycles_per_step'] = {f'wave{w}': {str(i): round(float(c)) for i, c in enumerate(phases[w]) if c > 0} for w in range(8) if phases[w].any()}
    if dry:
        out['data'] = 'dry-run (CPU stand-in for the device calls: not a measurement)'
        if scatter and world > 1:   # every rank's waveforms came back, and they are the ones of the clips it was sent
            want = all_mels.view(world, B, 80, T).to(torch.float64).sum(dim=(2, 3))
            ok = all(torch.equal(gathered[r][:, 0], want[r]) for r in range(world))
            out['dry_run_check'] = 'ok' if ok else 'MISMATCH'
    del model
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=0, help='ranks = GPUs of one node (default: WORLD_SIZE under torch.distributed.run, else 1)')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument('--frames', type=int, default=T_FRAMES)
    ap.add_argument('--batch', type=int, default=0, help='rows per GPU (default: the config\'s)')
    ap.add_argument('--kernel', default='auto', choices=['auto', 'team2', 'batch', 'batch_cs', 'simple'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the configs[2] / configs[4] legs of the default run')
    ap.add_argument('--phase-profile', action='store_true', help='run the instrumented loop kernel (wrnn_phase_profile) and attach cycles per phase')
    ap.add_argument('--dump', default=None, help='rank 0: save the waveforms of the last timed step (npz) -- used by tests/test_gpu_config3.py')
    ap.add_argument('--cpu-dry-run', action='store_true',
                    help='exercise the launcher / rendezvous / collectives / timing / JSON scaffolding on CPU (gloo) with a stand-in for the '
                         'device calls: tests only, the line it prints is not a measurement')
    args = ap.parse_args()

    under_launcher = 'RANK' in os.environ
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus == 0:
        args.gpus = world if under_launcher else 1
    if args.gpus > 1 and not under_launcher:
        return self_launch(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    dry = args.cpu_dry_run
    # configs[3] always runs its collectives, also on one rank: a standalone N=1 call provides its own rendezvous
    need_pg = world > 1 or args.config == 3
    if need_pg and not under_launcher:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            os.environ.setdefault('MASTER_PORT', str(s.getsockname()[1]))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if dry:
        dev = torch.device('cpu')
        if need_pg:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo')
    else:
        if not torch.cuda.is_available():
            print('bench.py: no HIP device visible (the hot path has no CPU fallback)', file=sys.stderr)
            return 3
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
        if need_pg:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('nccl', device_id=dev)

    copy_peak = None if dry or rank != 0 else measure_hbm_peaks(dev)
    out = run_config(args.config, world=world, rank=rank, dev=dev, dry=dry, steps=args.steps, warmup=args.warmup, frames=args.frames,
                     batch=args.batch, kernel_name=args.kernel, copy_peak=copy_peak, dump=args.dump, phase_profile=args.phase_profile)
    default_run = (args.config == 1 and world == 1 and not dry and not args.no_extra_configs and args.frames == T_FRAMES
                   and args.batch == 0 and args.kernel == 'auto')
    # the driver's SCALE command (`--gpus N`, config 1 by default) also reports BASELINE configs[3] at that N: 64 clips per GPU scattered
    # from / gathered on rank 0 over RCCL -- one line carries the weak B=1-per-GPU number and the throughput-mode number
    scale_extra = None
    if args.config == 1 and world > 1 and not args.no_extra_configs and args.batch == 0 and args.kernel == 'auto':
        # The headline is measured; this leg must not be able to take it down.  Its scatter / gather over RCCL has run on one rank (GPU box)
        # and on 2 / 8 gloo ranks (CPU) -- never on 8 real GPUs -- so a watchdog stands behind it: if the leg is not back within
        # SCALE_LEG_TIMEOUT seconds, rank 0 prints the headline line with the leg marked as timed out and every rank leaves.
        import threading
        printed = threading.Lock()   # the line is printed once: by the watchdog or by the main thread, whoever takes this first (never released)

        def give_up():
            if not printed.acquire(blocking=False):
                return                # the main thread is already printing: the leg came back right at the limit
            if rank == 0 and out is not None:
                out['extra_configs'] = {'3': {'error': f'configs[3] leg did not return within {SCALE_LEG_TIMEOUT} s (watchdog); headline unaffected'}}
                out['scale_leg_timed_out'] = True
                print(json.dumps(out), flush=True)
            os._exit(0 if rank == 0 else 75)   # rank 0 has delivered the headline; the other ranks leave with EX_TEMPFAIL so that the launcher's log shows the hang
        dog = threading.Timer(SCALE_LEG_TIMEOUT, give_up)
        dog.daemon = True
        dog.start()
        try:
            scale_extra = run_config(3, world=world, rank=rank, dev=dev, dry=dry, steps=2, warmup=1, frames=args.frames, batch=(2 if dry else 0),
                                     kernel_name='auto', copy_peak=None)
        except Exception as ex:   # every rank raises or none does (the collectives are symmetric); never sink the headline
            scale_extra = {'error': repr(ex)}
        dog.cancel()
        if not printed.acquire(blocking=False):   # the watchdog fired while the leg was returning: it prints and exits for us
            time.sleep(30)
            return 0
    if rank == 0 and out is not None:
        if scale_extra is not None:
            keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'roofline', 'config', 'data', 'dry_run_check', 'error')
            out['extra_configs'] = {'3': {k: scale_extra[k] for k in keep if k in scale_extra}}
        if default_run:
            # configs[2] and [4] in the driver's own run: a few steps each, attached to the headline line
            extra = {}
            note = lambda m: print(f'bench.py: [{time.strftime("%H:%M:%S")}] {m}', file=sys.stderr, flush=True)
            for cid in (2, 4):
                note(f'extra leg configs[{cid}]')
                try:
                    e = run_config(cid, world=1, rank=0, dev=dev, dry=False, steps=2, warmup=1, frames=T_FRAMES, batch=0, kernel_name='auto',
                                   copy_peak=copy_peak)
                    extra[str(cid)] = {k: e[k] for k in ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'roofline')}
                    extra[str(cid)]['config'] = e['config']
                except Exception as ex:  # an extra leg must never sink the headline
                    extra[str(cid)] = {'error': repr(ex)}
            for name, tgt in (('fold_auto', 'auto'), ('fold_hp_defaults', 11000), ('fold_per_xcd', 'per_xcd')):
                note(f'extra leg {name}')
                try:
                    extra[name] = fold_leg(dev, tgt)
                except Exception as ex:
                    extra[name] = {'error': repr(ex)}
            note('extra leg dm')
            try:
                extra['dm'] = dm_leg(dev)
            except Exception as ex:
                extra['dm'] = {'error': repr(ex)}
            note('extra leg train_step')
            try:
                extra['train_step'] = train_step_leg(dev)
            except Exception as ex:
                extra['train_step'] = {'error': repr(ex)}
            out['extra_configs'] = extra
        if world == 1 and not args.no_cpu_baseline and not dry:
            # cpu_baseline = the reference's own op sequence on torch-CPU on THIS box (the closest obtainable thing to "the reference CPU
            # wavernn_gen.py on the host cores of the same box": /root/reference does not exist here); cpu_port = the C restatement
            print('bench.py: cpu_baseline leg (torch CPU reference ops)', file=sys.stderr, flush=True)
            try:
                out['cpu_baseline'] = cpu_reference_ops()
            except Exception as e:  # the baseline must never sink the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'ksamples/s', 'cores': 0, 'kind': 'reference-ops (torch CPU, this box)',
                                       'sample': f'failed: {e!r}'}
            print('bench.py: cpu_port leg (C restatement)', file=sys.stderr, flush=True)
            try:
                out['cpu_port'] = cpu_baseline()
            except Exception as e:
                out['cpu_port'] = {'value': None, 'unit': 'ksamples/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {e!r}'}
            ref = cpu_reference()
            if ref:
                out['cpu_reference'] = ref
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
