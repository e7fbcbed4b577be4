#!/bin/bash
# Registers / scratch / LDS of every kernel in one .hip file, from hipcc's resource-usage remarks (no GPU needed):
#   tools/kernel_resources.sh tacotronv2_wavernn_chinese_amd/csrc/loop_batch.hip
# A non-zero ScratchSize inside a loop kernel means spills on the serial chain: look at it before going to the GPU.
f=${1:?usage: kernel_resources.sh file.hip}
cd "$(dirname "$f")" || exit 1
extra=""; [ "$(basename "$f")" = loop_batch_cs.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra -Wno-pass-failed -c "$(basename "$f")" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' |
  awk '/Name:/ {if (line) print line; line=$NF; next} {gsub(/^ +/,""); line=line " | " $0} END {print line}'
