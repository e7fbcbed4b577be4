"""Static census of loop_batch_kernel's generated ISA: register/spill report (-Rpass-analysis=kernel-resource-usage) and, per
barrier window (s_barrier to s_barrier) of each non-instrumented instantiation, instruction counts by class.

    python tools/static_census.py > profiles/rNN_static_census_batch.txt
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tacotronv2_wavernn_chinese_amd', 'csrc', 'loop_batch.hip')
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17']
NAMES = {'ILi0ELi2ELb0E': 'RAW_R8', 'ILi0ELi1ELb0E': 'RAW_R4', 'ILi1ELi2ELb0E': 'MOL_R8', 'ILi1ELi1ELb0E': 'MOL_R4'}


def main() -> int:
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'lb.s')
        r = subprocess.run([HIPCC, *FLAGS, '--cuda-device-only', '-S', SRC, '-o', asm, '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True, cwd=os.path.dirname(SRC))
        if r.returncode:
            sys.stderr.write(r.stderr)
            return 1
        print('# Static census of loop_batch_kernel as committed (hipcc -O3 --offload-arch=gfx950, ROCm 7.2; tools/static_census.py):')
        print('# -Rpass-analysis=kernel-resource-usage and, per barrier window of the generated ISA, instruction counts by class.')
        print('# Template arguments: <MODE (0 RAW, 1 MOL), NQ (row quads per team: 1 = 4 rows, 2 = 8 rows), PROF, PP (ping-pong schedule: opt-in)>.\n')
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r'remark: (.*?)\s*\[-Rpass', line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith('Function Name:'):
                if cur:
                    print(' \t'.join(cur))
                cur = [t.split(':', 1)[1].strip()] if 'Lb0ELb' in t else None
            elif cur is not None and re.match(r'(TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)', t):
                cur.append(t)
        if cur:
            print(' \t'.join(cur))
        txt = open(asm).read()
    for key, name in NAMES.items():
        m = re.search(r'\n_Z17loop_batch_kernel' + key + r'[^\n]*?:[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S)
        if not m:
            continue
        wins = [dict()]
        n = 0
        for l in m.group(1).split('\n'):
            l = l.strip()
            if not l or l[0] in ';.' or l.endswith(':'):
                continue
            n += 1
            op = l.split()[0]
            w = wins[-1]
            w['instr'] = w.get('instr', 0) + 1
            for cls, pred in (('mfma', op.startswith('v_mfma')), ('mfma_A_from_AGPR', op.startswith('v_mfma') and re.search(r'\], a\d+, ', l) is not None),
                              ('ds_read', op.startswith('ds_read')), ('ds_write', op.startswith('ds_write')),
                              ('buffer_load', op.startswith('buffer_load')), ('accvgpr_read', op.startswith('v_accvgpr_read')),
                              ('accvgpr_write', op.startswith('v_accvgpr_write')), ('waitcnt', op.startswith('s_waitcnt')),
                              ('s_nop', op.startswith('s_nop')), ('scratch_load', op.startswith('scratch_load')),
                              ('scratch_store', op.startswith('scratch_store'))):
                if pred:
                    w[cls] = w.get(cls, 0) + 1
            if op.startswith('s_barrier'):
                wins.append(dict())
        print(f'\n== {name}: {n} instructions; the windows holding MFMAs are the five barrier windows of a step, the ones before them the prologue')
        print('   (team formation, weight load / AGPR parking, row set-up)')
        cols = ['instr', 'mfma', 'mfma_A_from_AGPR', 'ds_read', 'ds_write', 'buffer_load', 'accvgpr_read', 'accvgpr_write', 'waitcnt', 's_nop',
                'scratch_load', 'scratch_store']
        for i, w in enumerate(wins):
            print(f'   window {i:2d}: ' + '  '.join(f'{c} {w.get(c, 0):4d}' for c in cols))
    return 0


if __name__ == '__main__':
    sys.exit(main())
