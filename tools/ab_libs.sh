#!/bin/bash
# A/B of library BUILDS on one box (run through gpurun from the repo root): one libfcn8s_hip.so per variant under scratch/lib_<name>.so (e.g. built with a
# -DFCN8S_LAB_... switch around the line in question), copied into place between alternating bench runs, so that box-to-box and warm-up drift cancel.
#   usage: tools/ab_libs.sh "<bench.py args>" <rounds> <name> <name> ...
# Prints per run: build, images/s, ms per step, the timed regions, and the kernel groups that usually move (ms per step, in-library HIP events).
ARGS="$1"; R=$2; shift 2
for i in $(seq 1 $R); do for l in "$@"; do
  cp scratch/lib_$l.so fcn8s_tensorflow_amd/libfcn8s_hip.so
  python bench.py --no-cpu-baseline --no-live-traffic $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['kernel_groups_ms_per_step']
print('%-8s'%'$l', d['value'], d['ms_per_step'], d['timed_regions_ms_per_step'], 'xform', g.get('wino_transform'), 'fwd', g.get('wino_gemm_fwd'), 'dgrad', g.get('wino_gemm_dgrad'), 'wgrad', g.get('wino_gemm_wgrad'))"
done; done
