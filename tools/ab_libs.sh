#!/bin/bash
# Same-box A/B of several builds of the library on one micro-benchmark:
#   tools/ab_libs.sh <outdir under gpurun_out> "<bench command>" name1 name2 ...   (name = suffix of csrc/libmfc_<name>.so; "new" = libmfc.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
CMD=$1; shift
mkdir -p $OUT
cd $R
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" == "new" ]; then lib=$R/meanflow_audio_codec_amd/csrc/libmfc.so; else lib=$R/meanflow_audio_codec_amd/csrc/libmfc_$n.so; fi
    MFC_LIB=$lib timeout -k 5 180 $CMD > $OUT/${n}_$rep.txt 2>&1 || { echo "FAILED $n rep $rep"; tail -5 $OUT/${n}_$rep.txt; exit 1; }
  done
done
for n in "$@"; do echo "== $n"; cat $OUT/${n}_2.txt; done
