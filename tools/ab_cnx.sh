#!/bin/bash
# A/B of two libmfc builds on the ConvNeXt micro-benchmark + SQ counters (one box, back to back)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2n
mkdir -p $OUT
cd $R
for rep in 1 2; do
  MFC_LIB=$R/meanflow_audio_codec_amd/csrc/libmfc_head.so timeout -k 5 120 python3 tools/bench_cnx.py 64 > $OUT/head_$rep.txt 2>&1 || exit 1
  timeout -k 5 120 python3 tools/bench_cnx.py 64 > $OUT/new_$rep.txt 2>&1 || exit 2
done
export TMPDIR=/tmp
cd /tmp
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
MFC_LIB=$R/meanflow_audio_codec_amd/csrc/libmfc_head.so timeout -k 5 200 rocprofv3 --pmc $CNT -d $OUT/pmc_head --output-format csv -- python3 $R/tools/bench_cnx.py 64 > $OUT/pmc_head.txt 2>&1 || exit 3
timeout -k 5 200 rocprofv3 --pmc $CNT -d $OUT/pmc_new --output-format csv -- python3 $R/tools/bench_cnx.py 64 > $OUT/pmc_new.txt 2>&1 || exit 4
paste $OUT/head_2.txt $OUT/new_2.txt
