"""Summary of tools/pmc_mfma.sh: per kernel (KITTI frame, Waymo-range frame, train step) the mean counter values per launch and
the derived matrix-pipe utilisation.  gpurun_out/pmcm_<run>_<COUNTER>.csv -> text + gpurun_out/pmc_mfma.json."""
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
KEEP = ("spconv_fwd_rows", "conv2d_bf16x3", "conv1x1", "dt_conv3", "dt_wgrad", "dt_dgrad", "spconv_fwd_wave", "spconv_bwd", "sa_mlp")
N_SIMD, N_XCD = 1024, 8
data = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))  # run -> kernel -> counter -> values
for f in sorted(glob.glob(os.path.join(root, "pmcm_*.csv"))):
    m = re.match(r"pmcm_([a-z]+)_(.+)\.csv", os.path.basename(f))
    run, counter = m.group(1), m.group(2)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if any(x in k for x in KEEP):
            data[run][k][counter].append(float(r["Counter_Value"]))
out = {}
for run, kernels in data.items():
    print(f"== {run}")
    for k, cs in sorted(kernels.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        mean = {c: sum(v) / len(v) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        gui = mean.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD  # summed over the XCDs -> cycles of the launch
        mf = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mops = mean.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + mean.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
        row = dict(launches=n, cycles_per_launch=gui, mfma_busy_cycles=mf, mfma_busy_frac_of_chip=(mf / (gui * N_SIMD) if gui else None),
                   mfma_instructions=mops / 32.0 if mops else None,  # MOPS counts 512-flop units: 32 per 16x16x32 instruction (r02 record)
                   busy_cycles=mean.get("SQ_BUSY_CYCLES"), waves=mean.get("SQ_WAVES"), wave_cycles_quad=mean.get("SQ_WAVE_CYCLES"),
                   fetch_bytes_x2=(2.0 * mean["FETCH_SIZE"] * 1024 if "FETCH_SIZE" in mean else None),  # KB, x2: the gfx950 correction
                   write_bytes=(mean["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in mean else None))
        out.setdefault(run, {})[k] = row
        if gui:
            print(f"  {k[:86]:86s} n={n:4d}  {gui / 2.4e3:7.2f} us at 2.4 GHz-equivalent cycles | MFMA busy {mf:12.0f} cycles = {100 * row['mfma_busy_frac_of_chip']:5.1f} % of 1 024 SIMDs"
                  f" | {row['mfma_instructions'] or 0:10.0f} MFMA instr | waves {row['waves'] or 0:7.0f} | fetch {((row['fetch_bytes_x2'] or 0) / 1e6):7.2f} MB write {((row['write_bytes'] or 0) / 1e6):7.2f} MB")
json.dump(out, open(os.path.join(root, "pmc_mfma.json"), "w"), indent=1)
