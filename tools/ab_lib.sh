# usage: bash tools/ab_lib.sh <name> <lib.so | default> <bench args...>: one bench line with an alternative build of the library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
name=$1; lib=$2; shift 2
if [ "$lib" != default ]; then export V3D_HIP_LIB=$lib; fi
timeout 600 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
python -c "import json,sys; d=json.loads(open('gpurun_out/ab/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), d.get('single_frame_ms'))"
