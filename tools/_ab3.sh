cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
timeout 1200 python -m pytest tests/test_gpu_plan.py tests/test_gpu_second.py tests/test_gpu_sparse_conv.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-fast-mode --no-h2d --no-extra --no-roofline --windows 15 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python -c "import json,sys; d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'frames/s', round(1e3*d['ms_per_step'],1), 'us/frame; single', round(1e3*d.get('single_frame_ms'),1), d['config'].get('pipeline_tuning'))"; }
run riders
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_noriders.so run noriders
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_prev.so run prev
run riders2
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_noriders.so run noriders2
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_prev.so run prev2
A="--pipeline 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-roofline --no-extra --windows 3 --steps 100"
bash tools/prof_stats.sh ab_riders_1f $A > gpurun_out/ab/riders_prof.txt 2>&1
