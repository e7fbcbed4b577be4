"""Static census of the step loops of loop_batch_cs_kernel's generated ISA (no GPU needed).

    python tools/loop_census.py [-DCS_KNOB=1 ...] > profiles/rNN_static_census_batch_cs.txt

For every non-instrumented instantiation: hipcc's resource remarks, and for each ROLE's step loop (the C waves' and the S waves' loops are separate
code: the innermost loops that contain workgroup barriers and MFMAs) the static instruction mix -- above all the spill traffic that sits on a step:
`scratch_load/store` (VGPR spills) and `v_readlane/v_writelane` (SGPR spills kept in VGPR lanes).  The pass set-up code around the two loops (row
tables, weights -> registers) is reported as "outside the step loops".
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tacotronv2_wavernn_chinese_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', '-Wno-pass-failed']
NAMES = {'ILi0ELi2ELb0E': 'RAW, 8 rows per team  <0,2,false>', 'ILi1ELi1ELb0E': 'MOL, 4 rows per team  <1,1,false>',
         'ILi0ELi1ELb0E': 'RAW, 4 rows per team  <0,1,false>', 'ILi1ELi2ELb0E': 'MOL, 8 rows per team  <1,2,false>'}
CLASSES = (('mfma', lambda op: op.startswith('v_mfma')), ('valu', lambda op: op.startswith('v_') and not op.startswith('v_mfma')),
           ('v_readlane', lambda op: op == 'v_readlane_b32'), ('v_writelane', lambda op: op == 'v_writelane_b32'),
           ('scratch_load', lambda op: op.startswith('scratch_load')), ('scratch_store', lambda op: op.startswith('scratch_store')),
           ('salu', lambda op: op.startswith('s_') and not op.startswith(('s_load', 's_nop', 's_waitcnt', 's_barrier', 's_sleep'))),
           ('s_nop', lambda op: op == 's_nop'), ('s_waitcnt', lambda op: op == 's_waitcnt'), ('ds', lambda op: op.startswith('ds_')),
           ('vmem', lambda op: op.startswith(('buffer_', 'global_', 'flat_'))), ('smem', lambda op: op.startswith('s_load')),
           ('s_barrier', lambda op: op == 's_barrier'))


def census(lines):
    c = {'instr': 0}
    for l in lines:
        s = l.strip()
        if not s or s[0] in ';.' or s.endswith(':'):
            continue
        op = s.split()[0]
        c['instr'] += 1
        for k, pred in CLASSES:
            if pred(op):
                c[k] = c.get(k, 0) + 1
    return c


# --windows: the VALU instructions of a step loop by barrier window and by what they are for (round-6 census for the "16-position mapping" question:
# which instructions are indexed by (unit, batch row) only -- 16 distinct values per wave at 4 rows per team, evaluated in all 64 lanes)
VALU_KINDS = (('transcendental (v_exp/v_rcp/v_log/v_sqrt/v_rsq)', lambda op: op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq'))),
              ('cross-lane (permlane / dpp / readlane / bpermute: K-phase folds, reductions)', lambda op, l='': op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'v_writelane')) or 'dpp' in op),
              ('compare / select (v_cmp, v_cndmask: tag checks, argmax, masks)', lambda op: op.startswith(('v_cmp', 'v_cndmask'))),
              ('fma / mul / add / sub (gates, folds, conditioning)', lambda op: op.startswith(('v_fma', 'v_mul', 'v_add', 'v_sub', 'v_pk_', 'v_mac', 'v_fmac'))),
              ('moves / conversions / bit ops', lambda op: True))


def valu_breakdown(lines):
    out = {}
    for l in lines:
        st = l.strip()
        if not st or st[0] in ';.' or st.endswith(':'):
            continue
        op = st.split()[0]
        if not op.startswith('v_') or op.startswith('v_mfma'):
            continue
        full = st.split(';')[0]
        for k, pred in VALU_KINDS:
            hit = pred(op) if k[0] != 'c' or not k.startswith('cross') else (op.startswith(('v_permlane', 'v_readlane', 'v_readfirstlane', 'v_writelane')) or '_dpp' in full or 'row_' in full or 'quad_perm' in full)
            if hit:
                out[k] = out.get(k, 0) + 1
                break
    return out


def windows(lines):
    """Split a step loop at its workgroup barriers: [(census, valu breakdown)] per window."""
    cuts = [i for i, l in enumerate(lines) if l.strip().startswith('s_barrier')]
    parts, a = [], 0
    for c in cuts + [len(lines)]:
        parts.append(lines[a:c])
        a = c + 1
    return [(census(p), valu_breakdown(p)) for p in parts if p]


def fmt(c):
    return ', '.join(f'{k} {c[k]}' for k in ['instr'] + [k for k, _ in CLASSES] if c.get(k))


def main() -> int:
    want_windows = '--windows' in sys.argv[1:]
    extra = [a for a in sys.argv[1:] if a.startswith('-') and a != '--windows']
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'cs.s')
        r = subprocess.run([HIPCC, *FLAGS, *extra, '--cuda-device-only', '-S', 'loop_batch_cs.hip', '-o', asm, '-Rpass-analysis=kernel-resource-usage'],
                           capture_output=True, text=True, cwd=CSRC)
        if r.returncode:
            sys.stderr.write(r.stderr)
            return 1
        txt = open(asm).read()
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:'):
            cur = t.split(':', 1)[1].strip()
            res[cur] = []
        elif cur and re.match(r'(TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill)', t):
            res[cur].append(t)
    print(f'# tools/loop_census.py {" ".join(extra)}: loop_batch_cs.hip as in the tree (hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form, ROCm 7.2)')
    for key, name in NAMES.items():
        fn = f'_Z20loop_batch_cs_kernel{key}Ev13WrnnBatchArgs'
        m = re.search(r'\n' + fn + r':[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S)
        if not m:
            continue
        lines = m.group(1).split('\n')
        print(f'\n== {name}: ' + ' | '.join(res.get(fn, [])))
        lab = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r'^(\.LBB\d+_\d+):', l))}
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r'\b(?:s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
            if mm and mm.group(1) in lab and lab[mm.group(1)] < i:
                loops.append((lab[mm.group(1)], i))
        # a role's step loop: the widest back-edge loop with >= 4 barriers that does not contain another such loop's MFMA count twice (i.e. not the pass loop)
        cand = [(a, b, census(lines[a:b + 1])) for a, b in loops]
        cand = [(a, b, c) for a, b, c in cand if c.get('s_barrier', 0) >= 4 and c.get('mfma', 0) >= 64]
        total_mfma = max((c['mfma'] for _, _, c in cand), default=0)
        roles = [(a, b, c) for a, b, c in cand if c['mfma'] < total_mfma]          # the pass loop holds both roles' MFMAs
        best = {}
        for a, b, c in roles:                                                       # widest loop per MFMA count (= per role)
            k = c['mfma']
            if k not in best or b - a > best[k][1] - best[k][0]:
                best[k] = (a, b, c)
        inside = 0
        for k, (a, b, c) in sorted(best.items(), key=lambda kv: kv[1][0]):
            role = 'S waves (shadow)' if a == min(v[0] for v in best.values()) else 'C waves (serial chain)'
            print(f'   step loop of the {role}: {fmt(c)}')
            sp = [l.strip().split(';')[0].strip() for l in lines[a:b + 1] if l.strip().startswith('scratch_')]
            if sp:
                print('      scratch traffic inside: ' + ' ; '.join(sp))
            inside += c['instr']
            if want_windows:
                tot = {}
                for wi, (wc, vb) in enumerate(windows(lines[a:b + 1])):
                    print(f'      window {wi}: instr {wc["instr"]}, mfma {wc.get("mfma", 0)}, valu {wc.get("valu", 0)}, salu {wc.get("salu", 0)}, ds {wc.get("ds", 0)}, vmem {wc.get("vmem", 0)}'
                          + (' | valu: ' + '; '.join(f'{v} {k.split(" (")[0]}' for k, v in vb.items()) if vb else ''))
                    for k, v in vb.items():
                        tot[k] = tot.get(k, 0) + v
                print('      VALU by kind over the step: ' + '; '.join(f'{v} {k}' for k, v in tot.items()))
        allc = census(lines)
        print(f'   whole kernel: {fmt(allc)}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
