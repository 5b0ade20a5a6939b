#!/bin/bash
# Collect the per-round evidence on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh r02_final
# -> gpurun_out/<tag>/: default bench line, rocprofv3 kernel stats of the same command, the FETCH_SIZE / WRITE_SIZE
#    PMC passes (separate runs, counters only) reduced by tools/pmc_traffic.py, and the decode path on its own.
# Every step is its own process and a failure stops the sequence (no GPU step runs after a failed one).
set -o pipefail
TAG=${1:-r02_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PMC_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-kernel-timing --no-overlap --no-loss-probe"
cd "$R" || exit 9
timeout -k 10 400 python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.log" || exit 1
echo "[profile] default bench done"
cd /tmp || exit 9
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-decode --no-loss-probe \
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
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
ls -la "$OUT"
