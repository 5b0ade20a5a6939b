#!/bin/bash
# SQ counters of the ConvNeXt kernels at the literal spatial size (GPU box, through gpurun):
#   tools/profile_cnx_sq.sh r03   ->  gpurun_out/r03_cnx_sq/r03_cnx_sq_counters.json  (copy it into profiles/)
# Counters only (no trace domains besides the kernel dispatch records rocprofv3 always writes); the program after `--`
# is python3 itself with an absolute script path (no env / shell hop: the profiler has initialised the GPU already).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_cnx_sq
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp || exit 9
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --pmc $CNT -d "$OUT/pmc" --output-format csv -- python3 "$R/tools/bench_cnx.py" 64 bf16 4 > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 "$R/tools/sq_counters.py" "$OUT/pmc" "$OUT/${TAG}_cnx_sq_counters.json" || exit 2
find "$OUT" -name "*.csv" -size +8M -delete
