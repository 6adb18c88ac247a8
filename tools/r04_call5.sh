#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
MSCL_LIB=$R/mscl_amd/csrc/build/libmscl_hip_stamp.so timeout -k 10 300 python -u tools/pp_stamps.py 2>&1 | grep -v amdgpu | tee $O/pp_stamps.log
