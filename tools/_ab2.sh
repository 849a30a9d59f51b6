cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
export V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_abl.so
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-fast-mode --no-h2d --no-extra --no-roofline --windows 9 > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python -c "import json,sys; d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1]); print('$name', round(1e3*d['ms_per_step'],1), 'us/frame', round(1e3*d.get('single_frame_ms'),1), d['config'].get('pipeline_tuning'))"; }
run all
V3D_ABL=r run no_ring64
V3D_ABL=c run no_conv
V3D_ABL=d run no_tile2d
V3D_ABL=cd run no_conv_tile2d
