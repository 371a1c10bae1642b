#!/bin/bash
# A/B of library options on the training bench (runs on the GPU box): tools/ab_bench.sh "<name>=<--option k=v ...>" ...
# prints images/s and ms/step per variant (10 steps, 3 warm-up, median of 3 timed regions).
for spec in "$@"; do
    name=${spec%%=*}; opts=${spec#*=}
    line=$(timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 3 $opts 2>/dev/null | tail -1)
    echo "$name: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step", d.get("timed_regions"))' 2>/dev/null || echo "failed: $line")"
done
