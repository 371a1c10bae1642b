#!/bin/bash
# Copies the judged summaries of one tools/collect_profiles.sh run (gpurun_out/prof_<tag>/) into profiles/ under round-1 names.
TAG=${1:?usage: publish_profiles.sh <tag>}
SRC=gpurun_out/prof_$TAG
DST=profiles
set -e
cp $SRC/bench_train_bs16.json            $DST/r01_bench_train_bs16.json
cp $SRC/bench_train_bs16_tf_adam.json    $DST/r01_bench_train_bs16_tf_adam.json
cp $SRC/bench_train_bs16_bf16_fc.json    $DST/r01_bench_train_bs16_bf16_fc.json
cp $SRC/bench_infer_bs1.json             $DST/r01_bench_infer_bs1.json
cp $SRC/bench_under_rocprof.json         $DST/r01_bench_under_rocprof.json
cp $SRC/stats/bench_kernel_stats.csv     $DST/r01_bench_train_kernel_stats.csv
cp $SRC/pmc_fetch/bench_counter_collection.csv $DST/r01_pmc_fetch_counter_collection.csv
cp $SRC/pmc_write/bench_counter_collection.csv $DST/r01_pmc_write_counter_collection.csv
cp $SRC/pmc_clock/bench_counter_collection.csv $DST/r01_pmc_clock_counter_collection.csv
cp $SRC/pmc_summary.txt                  $DST/r01_pmc_summary.txt
cp $SRC/pmc_clock_summary.txt            $DST/r01_pmc_clock_summary.txt
cp $SRC/pmc_clock.json                   $DST/r01_pmc_clock.json
cp $SRC/pmc_traffic.json                 $DST/pmc_traffic.json
echo "published $TAG"
