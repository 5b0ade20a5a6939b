#!/bin/bash
# Sweep the persistent grid of one ConvNeXt kernel kind (CnxKind index in csrc/convnext.hip) on the micro-benchmark:
#   tools/sweep_cnx_blocks.sh <outdir under gpurun_out> <kind> <pattern in bench_cnx output> n1 n2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; KIND=$2; PAT=$3; shift 3
mkdir -p $OUT
cd $R
for n in "$@"; do
  MFC_CNX_BLOCKS="$KIND:$n" timeout -k 5 180 python3 tools/bench_cnx.py 64 bf16 6 > $OUT/k${KIND}_$n.txt 2>&1 || { echo "FAILED $KIND $n"; tail -3 $OUT/k${KIND}_$n.txt; exit 1; }
  echo "kind $KIND blocks $n: $(grep "$PAT" $OUT/k${KIND}_$n.txt | tr -s ' ' | cut -d' ' -f1-8 | tr '\n' '|')"
done
