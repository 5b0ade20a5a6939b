#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked):
#   tools/collect_profiles.sh r03_final
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$TAG; D=$R/profiles
cp $S/bench_default.json $D/${TAG}_bench_default.json; cp $S/bench_default.log $D/${TAG}_bench_default.log
cp $S/bench_under_rocprof.json $D/${TAG}_bench_under_rocprof.json
cp $(ls $S/stats/*/*kernel_stats.csv | head -1) $D/${TAG}_rocprofv3_kernel_stats.csv
cp $S/${TAG}_pmc_traffic.json $D/${TAG}_pmc_traffic.json
cp $S/${TAG}_pmc_fetch_write_by_kernel.json $D/${TAG}_pmc_fetch_write_by_kernel.json
cp $(ls $S/decode_stats/*/*kernel_stats.csv | head -1) $D/${TAG}_decode_rocprofv3_kernel_stats.csv
for w in mnist_mlp mnist_mixer mdct; do cp $S/bench_$w.json $D/${TAG}_bench_$w.json; done
for w in mnist_mlp mnist_mixer; do cp $(ls $S/stats_$w/*/*kernel_stats.csv | head -1) $D/${TAG}_${w}_rocprofv3_kernel_stats.csv; done
cp $S/${TAG}_mixer_sq_counters.json $D/${TAG}_mixer_sq_counters.json
cp $R/gpurun_out/${TAG}_cnx_sq/${TAG}_cnx_sq_counters.json $D/${TAG}_cnx_sq_counters.json
cp $S/${TAG}_nstream_sq_counters.json $D/${TAG}_nstream_sq_counters.json; cp $S/${TAG}_nstream_lds_counters.txt $D/${TAG}_nstream_lds_counters.txt
ls -la $D | grep $TAG
