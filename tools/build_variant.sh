# usage: bash tools/build_variant.sh <suffix> "<extra hipcc flags>" [file.hip ...]  ->  vision3d_amd/lib/libvision3d_hip_<suffix>.so
# (then the default library is rebuilt).  Select with V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_<suffix>.so.  The files named
# (default: spconv.hip dense_train.hip) are the ones recompiled with the extra flags.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/vision3d_amd/csrc"
suffix=$1; flags=$2; shift 2 || true
files=${@:-"dense_train.hip spconv.hip"}
touch $files
make EXTRA="$flags" > /tmp/build_variant.log 2>&1 || { grep -E "error" -A6 /tmp/build_variant.log | head -30; exit 1; }
cp ../lib/libvision3d_hip.so ../lib/libvision3d_hip_$suffix.so
touch $files
make > /tmp/build_variant.log 2>&1 || { grep -E "error" -A6 /tmp/build_variant.log | head -30; exit 1; }
echo built $suffix
