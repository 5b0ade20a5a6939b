#!/bin/bash
# Build a variant of the library with extra -D flags on ONE source, for same-box A/B runs (MFC_LIB=<path> selects it):
#   tools/build_variant.sh exp1 convnext "-DMFC_CNX_EXP=1"   ->  meanflow_audio_codec_amd/csrc/libmfc_exp1.so (git-ignored)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/meanflow_audio_codec_amd/csrc
name=$1; src=$2; flags=$3
python -m meanflow_audio_codec_amd._build > /dev/null      # the other objects, up to date
mkdir -p /tmp/mfc_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $C/$src.hip -o /tmp/mfc_variants/${src}_$name.o -Wno-unused-function \
  -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $flags
objs=""
for o in $C/*.o; do
  b=$(basename $o .o)
  if [ "$b" == "$src" ]; then objs="$objs /tmp/mfc_variants/${src}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o $C/libmfc_$name.so $objs -L/opt/rocm/lib -lamdhip64
echo $C/libmfc_$name.so
