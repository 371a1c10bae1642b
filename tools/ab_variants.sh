#!/bin/bash
# Same-box A/B of compile-time variants of one kernel file: builds libfcn8s_hip.so once per value of a -D macro ON THE GPU BOX (the other objects travel with
# the snapshot), then alternates the libraries over ROUNDS rounds of one command.  Box-to-box spread of a bench line is 2-3 %: smaller effects need this.
#   usage: ab_variants.sh <file.hip> <MACRO> "<v1 v2 ...>" <rounds> <command...>
F=$1; MACRO=$2; VALS=$3; ROUNDS=$4; shift 4
cd fcn8s_tensorflow_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
cp ../libfcn8s_hip.so /tmp/lib_orig.so
B=${F%.hip}
for v in $VALS; do
  /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$v -c $F -o /tmp/${B}_$v.o || exit 1
  OBJS=""; for o in igemm gemm_bf16 elementwise winograd skinny model; do if [ $o = $B ]; then OBJS="$OBJS /tmp/${B}_$v.o"; else OBJS="$OBJS $o.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lib_$v.so $OBJS || exit 1
done
cd ../..
for r in $(seq 1 $ROUNDS); do for v in $VALS; do cp /tmp/lib_$v.so fcn8s_tensorflow_amd/libfcn8s_hip.so; echo -n "$MACRO=$v round $r: "; "$@"; done; done
cp /tmp/lib_orig.so fcn8s_tensorflow_amd/libfcn8s_hip.so
