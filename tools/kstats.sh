#!/bin/bash
# usage (on the GPU box, from the repo root): tools/kstats.sh <pattern-regex> [bench args...]
# rocprofv3 kernel stats of a short bench run, filtered to kernels whose name matches the pattern.
PAT="$1"; shift
export TMPDIR=/tmp
OUT=gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/bench.err
python - "$PAT" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[1])
f = glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if pat.search(r["Name"]):
        print("%-100s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
