#!/bin/bash
mkdir -p gpurun_out/r5k
export PYTHONPATH=/root/repo:$PYTHONPATH
for i in 1 2 3; do
  V3D_PRESPLIT=0 timeout -s USR1 --kill-after=8 100 python bench.py --no-cpu-baseline --no-fast-mode --no-h2d --windows 9 --steps 20 --warmup 5 --precision bf16x3 > gpurun_out/r5k/off_$i.json 2> gpurun_out/r5k/off_$i.err
  echo "run $i rc=$? bytes=$(stat -c %s gpurun_out/r5k/off_$i.json)"; grep -v amdgpu.ids gpurun_out/r5k/off_$i.err | tail -40
done
