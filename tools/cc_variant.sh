#!/bin/bash
# Compile one HIP source of the library with extra -D flags into /tmp and report registers / occupancy of its kernels:
#   tools/cc_variant.sh convnext "-DMFC_CNX_EXP=1" /tmp/isa/cnx_exp1.o
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $R/meanflow_audio_codec_amd/csrc/$1.hip -o $3 -Wall -Wno-unused-function \
  -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $2 -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|Occupancy|LDS Size|error|warning:" | paste - - - - | sed 's/remark: [^ ]* //g;s/\[-Rpass-analysis=kernel-resource-usage\]//g;s/'$1'.hip:[0-9]*:[0-9]*://g'
