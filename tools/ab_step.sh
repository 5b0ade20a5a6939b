#!/bin/bash
# Same-box A/B of several builds of the library on the WHOLE step (bench.py without the CPU legs), run in the order given
# (repeat names for an ABBA order):
#   tools/ab_step.sh <outdir under gpurun_out> new ct0 ct0 new     (name = suffix of csrc/libmfc_<name>.so; "new" = libmfc.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
i=0
for n in "$@"; do
  i=$((i + 1))
  if [ "$n" == "new" ]; then lib=$R/meanflow_audio_codec_amd/csrc/libmfc.so; else lib=$R/meanflow_audio_codec_amd/csrc/libmfc_$n.so; fi
  MFC_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-learn-probe > $OUT/${i}_$n.json 2> $OUT/${i}_$n.log || { echo "FAILED $n run $i"; tail -5 $OUT/${i}_$n.log; exit 1; }
  python3 -c "
import json; d=json.load(open('$OUT/${i}_$n.json')); print('$i $n', d['ms_per_step'], d['decode_ms_per_batch'], d['roofline']['frac'])"
done
