#!/bin/bash
# Collect the per-round evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh r03_final
# -> gpurun_out/<tag>/: default bench line, rocprofv3 kernel stats of the same command, the FETCH_SIZE / WRITE_SIZE
#    PMC passes (separate runs, counters only) reduced by tools/pmc_traffic.py, and the decode path on its own.
# Every step is its own process and a failure stops the sequence (no GPU step runs after a failed one).
set -o pipefail
TAG=${1:-r03_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-kernel-timing --no-overlap --no-loss-probe --no-learn-probe"
cd "$R" || exit 9
timeout -k 10 400 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.log" || exit 1
echo "[profile] default bench done"
cd /tmp || exit 9
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-decode --no-loss-probe --no-learn-probe \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.log" || exit 2
echo "[profile] kernel stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- python3 "$R/bench.py" $PMC_ARGS \
    > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.log" || exit 3
echo "[profile] FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" --output-format csv -- python3 "$R/bench.py" $PMC_ARGS \
    > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.log" || exit 4
echo "[profile] WRITE_SIZE pass done"
python3 "$R/tools/pmc_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/$TAG" > "$OUT/pmc_traffic.log" 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/decode_stats" -- python3 "$R/tools/bench_decode.py" 128 5 \
    > "$OUT/decode.json" 2> "$OUT/decode.log" || exit 6
echo "[profile] decode stats done"
cd "$R" || exit 9
# BASELINE configs #2 / #3 and the tokenizer alone (bench lines), then their kernel stats and the Mixer's SQ counters
for w in mnist_mlp mnist_mixer mdct; do
  timeout -k 10 500 python3 bench.py --workload $w --steps 20 --warmup 5 > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.log" || exit 7
done
echo "[profile] small workloads done"
cd /tmp || exit 9
for w in mnist_mlp mnist_mixer; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$w" -- python3 "$R/bench.py" --workload $w --steps 20 --warmup 5 --no-kernel-timing --no-cpu-baseline \
      > "$OUT/bench_${w}_under_rocprof.json" 2> "$OUT/bench_${w}_under_rocprof.log" || exit 8
done
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --pmc $CNT -d "$OUT/sq_mixer" --output-format csv -- python3 "$R/bench.py" --workload mnist_mixer --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline \
    > "$OUT/sq_mixer.json" 2> "$OUT/sq_mixer.log" || exit 10
python3 "$R/tools/sq_counters.py" "$OUT/sq_mixer" "$OUT/${TAG}_mixer_sq_counters.json" "^(gemm|adaln|gelu|colsum|transpose|chanmlp)" > "$OUT/sq_mixer_reduce.log" 2>&1 || exit 11
echo "[profile] small-workload profiles done"
# SQ counters (issue / wait split, LDS bank conflicts) of the N-streaming GEMM on its micro-benchmark
timeout -k 10 200 rocprofv3 --pmc $CNT -d "$OUT/sq_nstream" --output-format csv -- python3 "$R/tools/bench_nstream.py" > "$OUT/sq_nstream.log" 2>&1 || exit 13
python3 "$R/tools/sq_counters.py" "$OUT/sq_nstream" "$OUT/${TAG}_nstream_sq_counters.json" "^gemm_nstream" > "$OUT/sq_nstream_reduce.log" 2>&1 || exit 14
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR \
    -d "$OUT/sq_nstream_lds" --output-format csv -- python3 "$R/tools/bench_nstream.py" > "$OUT/sq_nstream_lds.log" 2>&1 || exit 15
python3 "$R/tools/pmc_sum.py" "$OUT/sq_nstream_lds" gemm_nstream > "$OUT/${TAG}_nstream_lds_counters.txt" 2>&1 || exit 16
echo "[profile] N-streaming counters done"
cd "$R" || exit 9
tools/profile_cnx_sq.sh "$TAG" > "$OUT/cnx_sq.log" 2>&1 || exit 12
find "$OUT" "$R/gpurun_out/${TAG}_cnx_sq" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" "$R/gpurun_out/${TAG}_cnx_sq" -name "*counter_collection.csv" -size +8M -delete
ls -la "$OUT"
