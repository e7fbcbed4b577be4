"""Developer probe (GPU box): cycles per phase marker of the batch / latency kernel (wrnn_phase_profile), one line per run.

    python tools/phase_profile.py 2 64 [frames [kernel]]     # config, rows per GPU  (frames 41, kernel auto)
"""
import json
import subprocess
import sys

cfg, batch = sys.argv[1], sys.argv[2]
frames = sys.argv[3] if len(sys.argv) > 3 else '41'
kernel = sys.argv[4] if len(sys.argv) > 4 else 'auto'
lib = sys.argv[5] if len(sys.argv) > 5 else None    # another build of the library (tools/build_variant.sh)
r = subprocess.run([sys.executable] + (['tools/ab_bench.py', lib] if lib else ['bench.py']) + ['--config', cfg, '--batch', batch, '--frames', frames, '--steps', '1', '--warmup', '1',
                    '--no-cpu-baseline', '--no-extra-configs', '--phase-profile', '--kernel', kernel], capture_output=True, text=True)
for l in r.stdout.splitlines():
    if l.startswith('{'):
        d = json.loads(l)
        for w, p in d.get('phase_cycles_per_step', {}).items():
            print(f'config {cfg} rows {batch} {kernel} {lib or ""} {w}: {p} total {sum(p.values())}')
        print('us/step', d['config']['us_per_step'])
if r.returncode:
    print(r.stderr[-2000:])
