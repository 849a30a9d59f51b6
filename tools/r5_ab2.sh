#!/bin/bash
mkdir -p gpurun_out/r5g
export PYTHONPATH=/root/repo:$PYTHONPATH
for v in "" _bf16mfma _bf16split _both ""; do
  V3D_HIP_LIB=/root/repo/vision3d_amd/lib/libvision3d_hip$v.so timeout 200 python tools/mb_prec_ab.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5g/prec_ab.txt | grep -E "lib:|64-> 64 K=27 n=  8160|32-> 32|sum:" | head -6
done
