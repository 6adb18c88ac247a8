#!/bin/bash
# Build libmscl_hip.so (gfx950) in-tree.  One translation unit per kernel family, compiled in parallel.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# NO packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) anywhere in the library: on gfx950 a v_pk_mul_f32 returns
# ZERO in the low half of its result for lanes 48-63 of a wave, rarely, while MFMA-heavy kernels of another hardware queue run on the same
# CU -- the three-stream training step is exactly that situation (root cause of the round-5 "dropped corner" of the trilinear
# up-sampling kernel; profiles/r06_flake.md; stand-alone reproducer tools/diag/flake_repro.hip, `--probe 0`).  hipcc emits these
# instructions on its own (SLP-vectorised float arithmetic: 4500 of them in this library before the switch), so the target feature is
# turned off for every translation unit.  (The host pass does not know the feature and says so: filtered below.)
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $NOPK"
mkdir -p build
pids=()
SRCS="conv_igemm conv_pp conv_dgrad_s2 conv_thin conv_k1 conv_halo conv_stem conv_wgrad conv_wgrad_halo conv_wgrad_stem bn_act elementwise pool3d color_aug datapath contrast optim"
for f in $SRCS; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ igemm.h -nt build/$f.o ] || [ ../../include/mscl_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o 2> >(grep -v "is not a recognized feature for this target" >&2) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""; for f in $SRCS; do OBJS="$OBJS build/$f.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libmscl_hip.so $OBJS
# the switch above must have taken: no packed fp32 instruction may be left in the device code of the library
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
if [ -x $OBJDUMP ]; then
  tmp=$(mktemp -d); cp libmscl_hip.so $tmp/ && ( cd $tmp && $OBJDUMP --offloading libmscl_hip.so >/dev/null 2>&1 )
  n=0
  for co in $tmp/libmscl_hip.so.*gfx950; do
    if [ -f "$co" ]; then c=$($OBJDUMP -d "$co" 2>/dev/null | grep -c "v_pk_[a-z]*_f32" || true); n=$((n + c)); fi
  done
  nco=$(ls $tmp/libmscl_hip.so.*gfx950 2>/dev/null | wc -l)
  rm -rf $tmp
  if [ "$nco" = "0" ]; then echo "warning: could not extract the device code objects to check them for packed fp32 instructions"; fi
  if [ "$n" != "0" ]; then echo "error: libmscl_hip.so holds $n packed fp32 instructions (see the NOPK note above)"; exit 1; fi
  echo "checked $nco code objects: no packed fp32 instructions"
fi
echo "built $(pwd)/libmscl_hip.so"
# a library with unresolved kernel stubs links fine but cannot be dlopen()ed: check now, not on the GPU box
python3 -c "import ctypes; ctypes.CDLL('$(pwd)/libmscl_hip.so')" || { echo "error: libmscl_hip.so has unresolved symbols (dlopen failed)"; exit 1; }
