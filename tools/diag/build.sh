#!/bin/bash
# Diagnostic builds for the dropped-load hunt (profiles/r06_flake.md): test infrastructure, not part of libmscl_hip.so.
#   libups_diag.so   the self-checking twin of the trilinear up-sampling kernel (upsample_diag.hip)
#   flake_repro      the stand-alone reproducer (flake_repro.hip: C ABI of libmscl_hip.so only, no PyTorch)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o libups_diag.so upsample_diag.hip
$HIPCC --offload-arch=gfx950 -O2 -std=c++17 -o flake_repro flake_repro.hip -ldl -lpthread
echo "built $(pwd)/libups_diag.so"
# `build.sh variants` also builds libmscl_hip_orig.so: the product library with the round-5 ORIGINAL up-sampling kernel (a global load per corner inside the loop;
# elementwise.hip of the commit before d4d050e), ten times the event rate of the shipped form: the discriminating runs use it through
# MSCL_LIB / --lib.  Needs the git history (built in the dev container; the .so travels to the GPU box).
if [ "$1" = "variants" ] && git -C ../.. rev-parse d4d050e^ >/dev/null 2>&1 && [ -f ../../mscl_amd/csrc/build/conv_pp.o ]; then
  mkdir -p build
  git -C ../.. show d4d050e^:mscl_amd/csrc/elementwise.hip > build/elementwise_orig.hip
  $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I../../mscl_amd/csrc -c build/elementwise_orig.hip -o build/elementwise_orig.o
  OBJS=""; for f in conv_igemm conv_pp conv_dgrad_s2 conv_thin conv_k1 conv_halo conv_stem conv_wgrad conv_wgrad_halo bn_act pool3d color_aug datapath contrast optim; do OBJS="$OBJS ../../mscl_amd/csrc/build/$f.o"; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o libmscl_hip_orig.so $OBJS build/elementwise_orig.o
  echo "built $(pwd)/libmscl_hip_orig.so"
  # the same kernel compiled WITHOUT packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32), two ways
  NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
  $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I../../mscl_amd/csrc $NOPK -c build/elementwise_orig.hip -o build/elementwise_orig_nopk.o 2> >(grep -v "is not a recognized feature" >&2)
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o libmscl_hip_orig_nopk.so $OBJS build/elementwise_orig_nopk.o
  # (round 6's shipped form WITH packed instructions -- the library itself is built without them now)
  $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -I../../mscl_amd/csrc -c ../../mscl_amd/csrc/elementwise.hip -o build/elementwise_pk.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o libmscl_hip_ship_pk.so $OBJS build/elementwise_pk.o
  echo "built the variants"
fi
