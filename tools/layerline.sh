python tools/layer_bench.py --precision bf16_train --batch 4 --height 1024 --width 2048 2>&1 | grep -E "fc6_fwd|fc6_dgrad|fc7_fwd|fc7_dgrad|total" | awk '{printf "%s %s | ", $1, $2}'; echo
