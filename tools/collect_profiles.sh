#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats and the two PMC passes of the same
# bench command.  Outputs under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_clock -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
python tools/pmc_clock_summary.py $OUT/pmc_clock/bench_counter_collection.csv $OUT/pmc_clock.json > $OUT/pmc_clock_summary.txt
python tools/pmc_summary.py $OUT/pmc_fetch/bench_counter_collection.csv $OUT/pmc_write/bench_counter_collection.csv $OUT/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline ($TAG)" > $OUT/pmc_summary.txt
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json      # the bench line below reads roofline.traffic from it (same box, same build, same command)
python bench.py --steps 20 --warmup 5 > $OUT/bench_train_bs16.json 2> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --mode infer --batch 1 --no-cpu-baseline > $OUT/bench_infer_bs1.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --precision bf16_fc --no-cpu-baseline > $OUT/bench_train_bs16_bf16_fc.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --optimizer adam --no-cpu-baseline > $OUT/bench_train_bs16_tf_adam.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --precision f32x3 --no-cpu-baseline > $OUT/bench_train_bs16_f32x3.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --precision f32x3 --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_f32x3.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --mode infer --batch 1 --precision f32x3 --no-cpu-baseline > $OUT/bench_infer_bs1_f32x3.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --mode infer --batch 1 --precision f32x2 --no-cpu-baseline > $OUT/bench_infer_bs1_f32x2.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --precision bf16_fwd --no-cpu-baseline > $OUT/bench_train_bs16_bf16_fwd.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --precision bf16_fwd --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_bf16_fwd.json 2>> $OUT/bench.err
for p in f32x2 bf16_fwd_x2 bf16_train; do
python bench.py --steps 10 --warmup 3 --precision $p --no-cpu-baseline > $OUT/bench_train_bs16_$p.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --precision $p --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_$p.json 2>> $OUT/bench.err
done
# BASELINE config 5's per-GPU shape and arithmetic (2048x1024, 4 images, bf16 fc6/fc7), config 2, and the end-to-end run
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --precision bf16_fc --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_bf16_fc.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_fp32.json 2>> $OUT/bench.err
python bench.py --mode e2e --steps 50 --warmup 5 --workers 32 > $OUT/bench_e2e_train_bs16.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --option deterministic=1 --no-cpu-baseline > $OUT/bench_train_bs16_deterministic.json 2>> $OUT/bench.err
# config 5's shape in bf16_train under rocprofv3 (kernel stats) and with the clock / matrix-pipe counters
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5_bf16_train -o bench -- python bench.py --steps 3 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_clock_c5_bf16_train -o bench -- python bench.py --steps 2 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
python tools/pmc_clock_summary.py $OUT/pmc_clock_c5_bf16_train/bench_counter_collection.csv $OUT/pmc_clock_c5_bf16_train.json > $OUT/pmc_clock_c5_bf16_train_summary.txt
python tools/layer_bench.py --precision bf16_train --batch 4 --height 1024 --width 2048 > $OUT/layer_bench_c5_bf16_train.txt 2>> $OUT/bench.err
python bench.py --gpus 2 --backend gloo --device 0 --steps 5 --warmup 2 --batch 8 --no-cpu-baseline > $OUT/bench_2ranks_one_gpu_gloo.json 2>> $OUT/bench.err
python tools/layer_bench.py > $OUT/layer_bench.txt 2>> $OUT/bench.err
python tools/layer_bench.py --infer --batch 1 --steps 10 > $OUT/layer_bench_infer_bs1.txt 2>> $OUT/bench.err
cat $OUT/bench_train_bs16.json
# where the waves of the bf16_train kernels spend their cycles (two more PMC passes of the same command; tools/pmc_wave_summary.py)
C5="python bench.py --steps 2 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline --no-secondary"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/wave_a -o bench -- $C5 > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/wave_b -o bench -- $C5 > /dev/null 2>> $OUT/bench.err
python tools/pmc_wave_summary.py bf16 $OUT/wave_a/bench_counter_collection.csv $OUT/wave_b/bench_counter_collection.csv > $OUT/c5_bf16_train_wave_state.txt
