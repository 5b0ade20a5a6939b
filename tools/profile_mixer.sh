#!/bin/bash
# Re-collect the Mixer part of a profile round (bench line, rocprofv3 kernel stats, SQ counters) after a change that only
# touches csrc/mixer.hip:   tools/profile_mixer.sh r03_final   (GPU box, through gpurun; same commands as tools/profile_round.sh)
set -o pipefail
TAG=${1:-r03_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_mixer
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R" || exit 9
timeout -k 10 500 python3 bench.py --workload mnist_mixer --steps 20 --warmup 5 > "$OUT/bench_mnist_mixer.json" 2> "$OUT/bench_mnist_mixer.log" || exit 1
cd /tmp || exit 9
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_mnist_mixer" -- python3 "$R/bench.py" --workload mnist_mixer --steps 20 --warmup 5 --no-kernel-timing --no-cpu-baseline \
    > "$OUT/bench_mnist_mixer_under_rocprof.json" 2> "$OUT/bench_mnist_mixer_under_rocprof.log" || exit 2
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --pmc $CNT -d "$OUT/sq_mixer" --output-format csv -- python3 "$R/bench.py" --workload mnist_mixer --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline \
    > "$OUT/sq_mixer.json" 2> "$OUT/sq_mixer.log" || exit 3
python3 "$R/tools/sq_counters.py" "$OUT/sq_mixer" "$OUT/${TAG}_mixer_sq_counters.json" "^(gemm|adaln|gelu|colsum|transpose|chanmlp)" > "$OUT/sq_mixer_reduce.log" 2>&1 || exit 4
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
ls "$OUT"
