#!/bin/bash
# static instruction census of the per-step loop of a RAW team kernel (developer tool; no GPU needed)
#   tools/count_instr.sh [loop_team2|loop_team]
set -e
K=${1:-loop_team2}
mkdir -p /tmp/w && cd /tmp/w
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only /root/repo/tacotronv2_wavernn_chinese_amd/csrc/$K.hip -o $K.s 2>/dev/null
python3 - "$K" <<'PY'
import re,collections,sys
K=sys.argv[1]
txt=open(f'/tmp/w/{K}.s').read()
m=re.search(r'^(_Z\d+%s_kernelILi0ELb0E\w*):' % K, txt, re.M)
start=m.start(); end=txt.index('s_endpgm', start)
lines=txt[start:end].split('\n')
labels={}
for i,l in enumerate(lines):
    mm=re.match(r'^(\.LBB\d+_\d+):',l)
    if mm: labels[mm.group(1)]=i
back=[]
for i,l in enumerate(lines):
    mm=re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)',l)
    if mm and mm.group(1) in labels and labels[mm.group(1)]<i: back.append((i-labels[mm.group(1)],labels[mm.group(1)],i))
back.sort(reverse=True)
mx=back[0][0]
span,lo,hi=[b for b in back if b[0]<0.9*mx][0]
body=[l.split()[0] for l in lines[lo:hi+1] if l.strip() and not l.strip().startswith(';') and not l.startswith('.')]
print(f'{m.group(1)}: step loop = asm lines {lo}-{hi}, {len(body)} instructions (static: both wave roles, all rarely-taken paths)')
cat=collections.Counter()
for op in body:
    op=re.sub(r'_e32|_e64','',op)
    if re.match(r's_(or|and|andn2|xor|mov)_b64|s_and_saveexec|s_or_saveexec|s_cbranch|s_branch',op): cat['execmask/branch']+=1
    elif op.startswith('s_waitcnt'): cat['waitcnt']+=1
    elif op.startswith('s_nop'): cat['nop']+=1
    elif op.startswith('s_barrier'): cat['barrier']+=1
    elif op.startswith('v_accvgpr'): cat['accvgpr']+=1
    elif re.match(r'v_pk_fma|v_pk_mul',op): cat['pk_fma/pk_mul']+=1
    elif re.match(r'v_fma|v_fmac',op): cat['fma']+=1
    elif re.match(r'v_exp|v_rcp|v_log|v_sqrt|v_rsq',op): cat['transcendental']+=1
    elif 'dpp' in op: cat['dpp']+=1
    elif op.startswith('v_mov'): cat['v_mov']+=1
    elif re.match(r'v_readlane|v_writelane|v_readfirstlane',op): cat['lane r/w (incl. sgpr spills)']+=1
    elif op.startswith('ds_'): cat['lds']+=1
    elif re.match(r'global_|scratch_',op): cat['vmem (global/scratch)']+=1
    elif op.startswith('s_'): cat['other salu']+=1
    else: cat['other valu']+=1
for k,v in cat.most_common(): print(f'  {k:30s} {v}')
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c /root/repo/tacotronv2_wavernn_chinese_amd/csrc/$K.hip -o /tmp/w/$K.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" | head -5
