export TMPDIR=/tmp
OUT=gpurun_out/prof_r05b
mkdir -p $OUT
python bench.py --steps 10 --warmup 3 --precision bf16_train --no-cpu-baseline > $OUT/bench_train_bs16_bf16_train.json 2>> $OUT/bench.err
python bench.py --steps 10 --warmup 3 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline > $OUT/bench_c5_2048x1024_bs4_bf16_train.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5_bf16_train -o bench -- python bench.py --steps 3 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_clock_c5_bf16_train -o bench -- python bench.py --steps 2 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline > /dev/null 2>> $OUT/bench.err
python tools/pmc_clock_summary.py $OUT/pmc_clock_c5_bf16_train/bench_counter_collection.csv $OUT/pmc_clock_c5_bf16_train.json > $OUT/pmc_clock_c5_bf16_train_summary.txt
python tools/layer_bench.py --precision bf16_train --batch 4 --height 1024 --width 2048 > $OUT/layer_bench_c5_bf16_train.txt 2>> $OUT/bench.err
python -c "
import json
for f in ('bench_train_bs16_bf16_train','bench_c5_2048x1024_bs4_bf16_train'):
    d=json.load(open('$OUT/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
"
head -8 $OUT/pmc_clock_c5_bf16_train_summary.txt
