# split-K / tile sweep of the small-map layers (needs the MSCL_IGEMM_KSPLIT / MSCL_IGEMM_CFG tuning hooks)
for shape in l3_256_256 l4_512_512 neck_333_p1 l3_128_256_s2 l4_256_512_s2; do
  for cfg in default 64,128,64 128,64,64; do
    for ks in 0 1 2 3 4 6 8 12 16; do
      if [ $cfg = default ]; then unset MSCL_IGEMM_CFG; else export MSCL_IGEMM_CFG=$cfg; fi
      if [ $ks = 0 ]; then unset MSCL_IGEMM_KSPLIT; else export MSCL_IGEMM_KSPLIT=$ks; fi
      echo "cfg=$cfg ks=$ks $(python tools/bench_conv.py --only $shape --modes fwd,dgrad --iters 20 2>&1 | grep -v amdgpu)"
    done
  done
done
