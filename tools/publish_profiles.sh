#!/bin/bash
# Copies the judged summaries of one tools/collect_profiles.sh run (gpurun_out/prof_<tag>/) into profiles/ under round names.
TAG=${1:?usage: publish_profiles.sh <tag> [round prefix, default r02]}
R=${2:-r06}
SRC=gpurun_out/prof_$TAG
DST=profiles
set -e
cp $SRC/bench_train_bs16.json            $DST/${R}_bench_train_bs16.json
cp $SRC/bench_train_bs16_tf_adam.json    $DST/${R}_bench_train_bs16_tf_adam.json
cp $SRC/bench_train_bs16_bf16_fc.json    $DST/${R}_bench_train_bs16_bf16_fc.json
cp $SRC/bench_infer_bs1.json             $DST/${R}_bench_infer_bs1.json
cp $SRC/bench_under_rocprof.json         $DST/${R}_bench_under_rocprof.json
cp $SRC/stats/bench_kernel_stats.csv     $DST/${R}_bench_train_kernel_stats.csv
cp $SRC/pmc_fetch/bench_counter_collection.csv $DST/${R}_pmc_fetch_counter_collection.csv
cp $SRC/pmc_write/bench_counter_collection.csv $DST/${R}_pmc_write_counter_collection.csv
cp $SRC/pmc_clock/bench_counter_collection.csv $DST/${R}_pmc_clock_counter_collection.csv
cp $SRC/pmc_summary.txt                  $DST/${R}_pmc_summary.txt
cp $SRC/pmc_clock_summary.txt            $DST/${R}_pmc_clock_summary.txt
cp $SRC/pmc_clock.json                   $DST/${R}_pmc_clock.json
cp $SRC/pmc_traffic.json                 $DST/pmc_traffic.json
for f in bench_c5_2048x1024_bs4_bf16_fc bench_c5_2048x1024_bs4_fp32 bench_e2e_train_bs16 bench_2ranks_one_gpu_gloo bench_train_bs16_f32x3 bench_c5_2048x1024_bs4_f32x3 bench_infer_bs1_f32x3 bench_train_bs16_bf16_fwd bench_c5_2048x1024_bs4_bf16_fwd bench_train_bs16_f32x2 bench_c5_2048x1024_bs4_f32x2 bench_train_bs16_bf16_fwd_x2 bench_c5_2048x1024_bs4_bf16_fwd_x2 bench_infer_bs1_f32x2 bench_train_bs16_bf16_train bench_c5_2048x1024_bs4_bf16_train bench_train_bs16_deterministic; do
    [ -f $SRC/$f.json ] && cp $SRC/$f.json $DST/${R}_$f.json
done
[ -f $SRC/layer_bench.txt ] && cp $SRC/layer_bench.txt $DST/${R}_layer_bench.txt
[ -f $SRC/layer_bench_infer_bs1.txt ] && cp $SRC/layer_bench_infer_bs1.txt $DST/${R}_layer_bench_infer_bs1.txt
python tools/roofline_table.py $DST/${R}_bench_train_bs16.json > $DST/${R}_roofline_table.md
[ -f $DST/${R}_bench_c5_2048x1024_bs4_bf16_fwd.json ] && python tools/roofline_table.py $DST/${R}_bench_c5_2048x1024_bs4_bf16_fwd.json > $DST/${R}_roofline_table_c5_bf16_fwd.md
[ -f $DST/${R}_bench_c5_2048x1024_bs4_bf16_train.json ] && python tools/roofline_table.py $DST/${R}_bench_c5_2048x1024_bs4_bf16_train.json > $DST/${R}_roofline_table_c5_bf16_train.md
[ -f $SRC/stats_c5_bf16_train/bench_kernel_stats.csv ] && cp $SRC/stats_c5_bf16_train/bench_kernel_stats.csv $DST/${R}_c5_bf16_train_kernel_stats.csv
[ -f $SRC/pmc_clock_c5_bf16_train_summary.txt ] && cp $SRC/pmc_clock_c5_bf16_train_summary.txt $DST/${R}_c5_bf16_train_pmc_clock_summary.txt
[ -f $SRC/pmc_clock_c5_bf16_train.json ] && cp $SRC/pmc_clock_c5_bf16_train.json $DST/${R}_c5_bf16_train_pmc_clock.json
[ -f $SRC/layer_bench_c5_bf16_train.txt ] && cp $SRC/layer_bench_c5_bf16_train.txt $DST/${R}_layer_bench_c5_bf16_train.txt
[ -f $SRC/c5_bf16_train_wave_state.txt ] && cp $SRC/c5_bf16_train_wave_state.txt $DST/${R}_c5_bf16_train_wave_state.txt
echo "published $TAG as $R"
