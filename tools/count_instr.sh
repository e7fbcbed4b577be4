#!/bin/bash
# static instruction census of the per-step loop of the RAW team kernel (developer tool; no GPU needed)
set -e
cd /tmp/w
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only /root/repo/tacotronv2_wavernn_chinese_amd/csrc/loop_team.hip -o loop_team.s 2>/dev/null
awk '/_Z16loop_team_kernelILi0ELb0EEv12WrnnTeamArgs:/,/s_endpgm/' loop_team.s > raw.s
python3 - <<'PY'
import re,collections
lines=open('/tmp/w/raw.s').read().split('\n')
labels={}
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB0_\d+):',l)
    if m: labels[m.group(1)]=i
back=[]
for i,l in enumerate(lines):
    m=re.match(r'\s+s_c?branch\S*\s+(\.LBB0_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i: back.append((i-labels[m.group(1)],labels[m.group(1)],i))
back.sort(reverse=True)
mx=back[0][0]
span,lo,hi=[b for b in back if b[0]<0.9*mx][0]
body=[l.split()[0] for l in lines[lo:hi+1] if l.strip() and not l.strip().startswith(';') and not l.startswith('.')]
print(f'step loop: asm lines {lo}-{hi}, {len(body)} instructions (static, includes rarely-taken paths)')
cat=collections.Counter()
for op in body:
    op=re.sub(r'_e32|_e64','',op)
    if re.match(r's_(or|and|andn2|xor|mov)_b64|s_and_saveexec|s_or_saveexec|s_cbranch|s_branch',op): cat['execmask/branch']+=1
    elif op.startswith('s_waitcnt'): cat['waitcnt']+=1
    elif op.startswith('s_nop'): cat['nop']+=1
    elif op.startswith('v_accvgpr_read'): cat['accvgpr_read']+=1
    elif re.match(r'v_pk_fma|v_fma|v_fmac',op): cat['fma']+=1
    elif op.startswith('v_mov'): cat['v_mov']+=1
    elif re.match(r'v_readlane|v_writelane',op): cat['lane r/w (sgpr spill)']+=1
    elif op.startswith('ds_'): cat['lds']+=1
    elif re.match(r'global_|scratch_',op): cat['vmem']+=1
    elif op.startswith('s_'): cat['other salu']+=1
    else: cat['other valu']+=1
for k,v in cat.most_common(): print(f'  {k:24s} {v}')
PY
grep -E "VGPRs Spill|SGPRs Spill" <(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c /root/repo/tacotronv2_wavernn_chinese_amd/csrc/loop_team.hip -o /tmp/w/lt.o -Rpass-analysis=kernel-resource-usage 2>&1) | head -2
