cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-fast-mode --no-h2d --no-extra --no-roofline > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python -c "import json,sys; d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), 'frames/s', round(1e3*d['ms_per_step'],1), 'us/step; single', round(1e3*d.get('single_frame_ms'),1), d['config'].get('pipeline_tuning'))"; }
for i in 1 2; do
run waymo_riders$i --workload waymo --windows 9 --steps 60
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_noriders.so run waymo_noriders$i --workload waymo --windows 9 --steps 60
run bs8_riders$i --batch 8 --windows 9 --steps 30
V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_noriders.so run bs8_noriders$i --batch 8 --windows 9 --steps 30
done
