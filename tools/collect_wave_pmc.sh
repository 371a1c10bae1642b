# wave-state counters of the bf16_train step at config 5's shape (two PMC passes, kernel trace only)
export TMPDIR=/tmp
OUT=gpurun_out/prof_wave
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --height 1024 --width 2048 --batch 4 --precision bf16_train --no-cpu-baseline --no-secondary"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o bench -- $CMD > /dev/null 2>> $OUT/err.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/b -o bench -- $CMD > /dev/null 2>> $OUT/err.log
A=$(find $OUT/a -name "*counter_collection.csv" | head -1); B=$(find $OUT/b -name "*counter_collection.csv" | head -1)
python tools/pmc_wave_summary.py bf16 $A $B | tee $OUT/wave_summary.txt
tail -3 $OUT/err.log
