# In-frame ablation of the register-gather ring kernel (one frame at a time, rocprofv3 kernel statistics): 0 = as is, 1 = no gathers,
# 2 = no weight stream, 3 = neither (results wrong by construction; needs the -DV3D_EXPERIMENTS build: tools/build_variant.sh exp -DV3D_EXPERIMENTS)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export V3D_HIP_LIB=vision3d_amd/lib/libvision3d_hip_exp.so
for d in 0 1 2 3; do
  rm -rf /tmp/pd$d
  V3D_RING_REGS=1 V3D_RING_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd$d -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --pipeline 1 > /tmp/pd$d.json 2>/tmp/pd$d.err
  f=$(find /tmp/pd$d -name "*kernel_stats.csv" | head -1)
  python - $f $d <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "spconv_fwd_rows_ring<64, 64" in r["Name"]:
        print("V3D_RING_DBG=%s  %-60s calls %6s avg %6.2f us" % (sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
