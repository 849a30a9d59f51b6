# usage: bash tools/build_variant.sh <suffix> "<extra hipcc flags>"  ->  vision3d_amd/lib/libvision3d_hip_<suffix>.so (then the
# default library is rebuilt).  Select with V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_<suffix>.so
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/vision3d_amd/csrc"
touch dense_train.hip spconv.hip
make EXTRA="$2" > /tmp/build_variant.log 2>&1 || { grep -E "error" -A6 /tmp/build_variant.log | head -30; exit 1; }
cp ../lib/libvision3d_hip.so ../lib/libvision3d_hip_$1.so
touch dense_train.hip spconv.hip
make > /tmp/build_variant.log 2>&1 || { grep -E "error" -A6 /tmp/build_variant.log | head -30; exit 1; }
echo built $1
