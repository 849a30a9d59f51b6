"""bench.py -- frames/sec of the SECOND forward on synthetic 16k-point KITTI-range clouds (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: bench.py starts the N ranks itself, launch_plan)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch (bs=1/GPU, configs[1]): device voxelizer + VFE -> 14-layer sparse 3-D conv
backbone -> .dense() BEV -> dense RPN (MFMA, background tiles skipped) -> proposal stage (top-k, decode, rotated NMS, score
cut), all hand-written HIP, replayed as one HIP graph (37 kernels) with a single 8-byte host read (proposal count + the plan's
capacity-overflow word).  The timed loop cycles through --stream (default 8) DIFFERENT clouds per rank, resident in HBM before
the timed region.  By default several frames are in flight per GPU (one graph and plan arena per slot, one slot more than
frames in flight; up to 4, streams and depth picked by measurement before the warm-up -- config.pipeline_tuning); every step
submits one frame and the timed region completes exactly K of them.  The same graph run one frame at a time is reported as
single_frame_ms / frames_per_s_one_at_a_time.  Frames are independent, so N GPUs run N replicas on different frames with no
data-path collective (weak scaling); the only communication is the timing barrier / max-reduce.  Rank 0 prints ONE JSON line.
Other lines of BASELINE.json: --workload waymo (configs[4]), --mode train (configs[2]), --mode pvrcnn (configs[3]).

Extra objects on that line:
  roofline        the dominant sparse kernel (KITTI: spconv_fwd_rows_ring<64,64>, 7 launches/frame; Waymo:
                  spconv_fwd_rows_kouter<64,64>): algorithmic bytes A_min = 4*(N_in*Cin + N_out*Cout + K*Cin*Cout) + 8*R per launch
                   divided by its average duration measured with HIP events over graph-captured repeats on the launch
                  stream, vs 8 TB/s; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/).
  roofline_dense  the 3x3 RPN convolution against the bf16 MFMA peak (every tile convolved, random input).
  cpu_baseline    oracle/ (scalar C sparse path + torch CPU dense path) timed on the host on a bounded sample of the same
                  workload -- N=1, rank 0 only: 1 thread, and `all_cores` = one frame per single-thread worker on ALL PHYSICAL cores;
                  `cpu_model`, `host_cores`, `host_threads` name the box.  Baseline only.
  value_p10/p90   spread of the per-window throughput over `windows` back-to-back K-step windows (`value` = the median window);
  with_h2d        the same windows with every step's cloud copied in from pinned host memory (SURVEY 8d timing variant).
"""
import argparse
import json
import os
import sys
import time

# more hardware queues for the runtime to spread streams over (the pipeline picks its streams by measurement, see
# vision3d_amd/detector/graph.py:PipelinedSecond.tune); must be set before the HIP runtime starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# RCCL ("nccl") over xGMI is the launch contract.  V3D_BENCH_BACKEND=gloo exists only to exercise the multi-rank control flow
# (env parsing, barrier, max-over-ranks, rank-0 JSON) where there are fewer GPUs than ranks: ranks then share devices.
BACKEND = os.environ.get("V3D_BENCH_BACKEND", "nccl")
REDUCE_DEVICE = "cuda" if BACKEND == "nccl" else "cpu"
MAX_PIPELINE = int(os.environ.get("V3D_BENCH_MAX_PIPELINE", "4"))  # slots the autotuned pipeline may use


DTYPE_NOTE = {
    "fp32": "f16x3-scaled (fp32-class): fp32 operands split into f16 hi + lo pieces of x * s under per-tensor power-of-two scales, "
            "3 MFMA terms (lo*hi + hi*lo + hi*hi), fp32 accumulate; relative product error 2^-22 -- the result differs from the "
            "reference's fp32 modules by fp32 summation noise: strict elementwise relative error against float64 <= 2e-4 on entries "
            "above 1e-3 of a layer's maximum, where torch's own fp32 conv3d shows up to 1.1e-4 (tests/test_gpu_conv3d_parity.py); end to "
            "end against float64 (tests/test_gpu_second.py, oracle second_forward64): BEV map 3.4e-5 (the fp32 CPU restatement: 4.7e-5), "
            "P_reg <= 1.2e-4 (0.7e-4), P_cls 8e-8, the intermediate RPN map <= 4.9e-4 (2.0e-4) -- noisier than torch's fp32 modules by "
            "up to 2.5x in the dense head; a frame whose tensors leave the calibrated range of the scales in either direction is flagged, "
            "recalibrated on and run again",
    "bf16x3": "bf16x3: fp32 operands split into bf16 hi + lo, 3 MFMA terms (lo*hi + hi*lo + hi*hi), fp32 accumulate -- a 16 x 16-bit "
              "split product, relative product error 2^-17 (NOT fp32: 2^-24); strict elementwise relative error <= 3e-3 on entries "
              "above 1e-3 of a layer's maximum (tests/test_gpu_conv3d_parity.py)",
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks of the job = GPUs of this node, one process per GPU.  Under torch.distributed.run (WORLD_SIZE set) it "
                         "must equal WORLD_SIZE; without a launcher and N > 1 bench.py starts the N ranks itself (launch_plan)")
    ap.add_argument("--steps", type=int, default=300, help="timed steps (a 50-step window is 16 ms at KITTI size: too short to be stable)")
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step (configs[1] = 1)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--workload", choices=["kitti", "waymo"], default="kitti",
                    help="kitti: BASELINE configs[1] (16k-pt KITTI-range cloud, the headline metric); "
                         "waymo: configs[4] (180k-pt sweep, +-75.2 m, 0.05 m voxels: the HBM stress case)")
    ap.add_argument("--path", choices=["graph", "native", "eager"], default="graph",
                    help="graph: native path captured in one HIP graph; native: backbone plan + MFMA dense head, eager "
                         "launches; eager: per-op python -> C ABI (every path runs the hand-written kernels: there is no torch dense path)")
    ap.add_argument("--mode", choices=["forward", "train", "pvrcnn", "plumbing"], default="forward",
                    help="forward: the headline metric; train: BASELINE configs[2] (SECOND train step, bs=8/GPU, gradient "
                         "all-reduce over RCCL) -- a secondary line, same JSON contract")
    ap.add_argument("--no-channels-last", action="store_true", help="train mode: keep the dense RPN/head in NCHW")
    ap.add_argument("--train-arith", choices=["fp32", "bf16"], default="bf16",
                    help="train mode, arithmetic of `value`: bf16 = the step under bf16 autocast (BASELINE configs[2]: \"train step bf16\"), "
                         "fp32 = the reference's train.py as written (no autocast; dense half on the fp32-class split kernels).  The other "
                         "one is measured too and printed under `fp32_script` / `fast_mode` (--no-fast-mode: not at all)")
    ap.add_argument("--no-amp", action="store_true", help="train mode: same as --train-arith fp32 (kept for older command lines)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="frames in flight per GPU (graph path).  Throughput mode: independent bs=1 frames overlap on separate "
                         "streams / HIP graphs / plan arenas; every step still submits ONE frame and the timed region completes "
                         "exactly K frames.  0 (default): up to 4 in flight, streams and depth picked by measurement before the "
                         "warm-up (PipelinedSecond.tune); N >= 2: exactly N on streams in creation order; 1: one frame at a time "
                         "(always reported beside it as single_frame_ms / frames_per_s_one_at_a_time)")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default="fp32",
                    help="arithmetic of the native inference path: fp32 = f16 hi/lo pieces under calibrated power-of-two scales (the "
                         "reference's fp32 modules up to summation noise); bf16x3 = bf16 pieces, 2^-17 per product (fast mode)")
    ap.add_argument("--no-fast-mode", action="store_true", help="forward mode: skip the bf16x3 line measured beside the fp32-class one")
    ap.add_argument("--order", choices=["shuffled", "scan", "morton"], default="shuffled",
                    help="forward mode: point order of the synthetic sweeps -- shuffled (default: what the TRAINING dataset hands over, "
                         "kitti_dataset.py:154; the gathers of the sparse convolutions are then random) or scan (firing order: what "
                         "inference.py reads from a .bin file; neighbouring voxels are neighbouring rows)")
    ap.add_argument("--watchdog", type=int, default=1500, help="seconds after which a run that has not finished dumps its stacks and exits 124 (0 = off)")
    ap.add_argument("--stream", type=int, default=8, help="different synthetic frames per rank the timed loop cycles through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch event timing pass (profiling runs)")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU baseline sample (about 1.4 s each on one core: ~6 s)")
    ap.add_argument("--windows", type=int, default=31,
                    help="forward mode: back-to-back timed windows of --steps steps each (barrier + synchronize on both sides of every "
                         "window, pipeline empty at its start, max over ranks per window); `value` is the MEDIAN window, "
                         "value_p10 / value_p90 the spread.  The count is cut down so that the windows take at most ~20 s")
    ap.add_argument("--end-to-end", action="store_true",
                    help="--mode pvrcnn: time PV_RCNN.inference(item) from raw points (device voxelizer + sparse CNN + stage-1 head + "
                         "stage 2 + refinement NMS), one frame at a time, instead of stage 2 on resident stage-1 outputs")
    ap.add_argument("--no-h2d", action="store_true", help="skip the with_h2d line (pinned host cloud copied in every step)")
    ap.add_argument("--single-frames", type=int, default=200, help="frames timed one at a time for single_frame_ms (median, p10, p90)")
    ap.add_argument("--no-extra", action="store_true",
                    help="default forward run on one GPU: skip the compact sub-lines of the OTHER BASELINE configurations (`extra`: Waymo-range "
                         "sweep, bs = 8 KITTI batch, plumbing, PV-RCNN stage 2, train step), each a short in-process run of its own mode")
    return ap.parse_args(argv)


def launch_plan(gpus, environ, argv, n_devices, backend=BACKEND, port=None):
    """How `--gpus N` becomes N ranks (the reference has no multi-GPU mode at all, training.md:6: this axis is the repository's).

    Returns None when this process IS a rank of the right job (WORLD_SIZE == N, or N == 1 without a launcher), or the argv of
    the launcher to re-execute under: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>` -- the form the driver uses itself.  Raises instead of quietly running a
    different job: WORLD_SIZE != N, fewer GPUs than ranks over RCCL (gloo, the control-flow check, lets ranks share devices)."""
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus}: need at least one rank")
    world = environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report "
                             f"a {world}-rank job as n_gpus={gpus}")
        return None
    if backend == "nccl" and n_devices < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but this node shows {n_devices} GPU(s): one process per GPU over RCCL needs {gpus} "
                         "(V3D_BENCH_BACKEND=gloo runs the N-rank control flow on fewer devices)")
    if gpus == 1:
        return None
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def ensure_world(args):
    """Called first by main(): re-executes under the launcher when --gpus N > 1 came without one (never returns then)."""
    cmd = launch_plan(args.gpus, os.environ, sys.argv[1:], torch.cuda.device_count())
    if cmd is None:
        return
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def numa_cpus_of_gpu(device_index, sysfs="/sys"):
    """CPUs of the NUMA node the GPU hangs off (sorted list), or None when the topology cannot be read: the device's PCI address
    from torch -> /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/node<N>/cpulist."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read().strip())
        if node < 0:
            return None
        return parse_cpulist(open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")).read())
    except Exception:
        return None


def parse_cpulist(text):
    """'0-15,128-143' -> [0, ..., 15, 128, ..., 143]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return sorted(set(cpus))


def pin_rank_to_gpu_numa(local_rank, world):
    """One process per GPU: the frame is ~37 launches of ~10 us driven by ONE host thread, so where that thread runs matters once
    eight ranks share the host -- each rank is pinned to the cores of its GPU's NUMA node (its share of them when several local
    ranks hang off one node).  Multi-rank jobs only (V3D_BENCH_PIN=1 / 0 forces it on / off); returns what was done for the line."""
    want = os.environ.get("V3D_BENCH_PIN")
    if want == "0" or (want is None and world <= 1):
        return None
    cpus = numa_cpus_of_gpu(local_rank % max(torch.cuda.device_count(), 1))
    if not cpus:
        return dict(pinned=False, reason="GPU -> NUMA node not readable from sysfs")
    try:
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0))) or cpus
        os.sched_setaffinity(0, allowed)
        return dict(pinned=True, cpus=len(allowed), first=allowed[0], last=allowed[-1])
    except Exception as e:
        return dict(pinned=False, reason=f"{type(e).__name__}: {str(e)[:80]}")


def ranks_seen(world, expect=None):
    """An all-reduce of ones over the job: what the collective backend (RCCL) really spans.  `expect` (= --gpus): a job whose
    collective spans a different number of ranks fails here, before anything is timed."""
    seen = 1
    if world > 1:
        import torch.distributed as dist
        ones = torch.ones(1, device=REDUCE_DEVICE)
        dist.all_reduce(ones)
        seen = int(ones.item())
    if expect is not None and seen != expect:
        raise SystemExit(f"bench.py: --gpus {expect} but the collective spans {seen} rank(s)")
    return seen


def layer_algorithmic_bytes(stats):
    """A_min of one sparse layer, fp32."""
    return 4 * (stats["n_in"] * stats["cin"] + stats["n_out"] * stats["cout"] + stats["K"] * stats["cin"] * stats["cout"]) \
        + 8 * stats["pairs"]


def mfma_busy_from_profiles(run, kernel_substr, avg_us):
    """Matrix-pipe occupancy of a kernel from the committed counter passes (profiles/pmc_mfma.json, tools/pmc_mfma.sh):
    SQ_VALU_MFMA_BUSY_CYCLES per launch / (1 024 SIMDs x THIS run's launch duration x 2.4 GHz) -- counters cannot be read from inside
    this process, and the durations under counter collection are inflated.  None when no counter record matches."""
    path = os.path.join(REPO, "profiles", "pmc_mfma.json")
    if not os.path.exists(path) or not avg_us:
        return None
    rec = json.load(open(path))
    hits = [(k, v) for k, v in rec.get(run, {}).items() if kernel_substr in k]
    if not hits:
        return None
    n = sum(v["launches"] for _, v in hits)
    busy = sum(v["mfma_busy_cycles"] * v["launches"] for _, v in hits) / n
    instr = sum((v.get("mfma_instructions") or 0.0) * v["launches"] for _, v in hits) / n
    return dict(mfma_busy_cycles_per_launch=busy, mfma_instructions_per_launch=instr, launches_in_record=n,
                mfma_busy_frac=busy / (1024.0 * avg_us * 1e-6 * 2.4e9), clock_assumed_ghz=2.4,
                sustained_floor_us=instr / 1024.0 * 10e-3,  # 10 ns per instruction and SIMD on random operands (profiles/r06_mfma_chain.txt)
                source=rec.get("source"))


def train_main(args):
    """configs[2]: one optimiser step of SECOND per GPU batch; gradients all-reduced over RCCL in two buckets, the dense half's
    while the native sparse backward runs (dist_util.TwoPhaseGradReducer).

    step = device voxelizer -> sparse backbone (training plan: HIP kernels forward and backward) -> dense RPN/head (hand-written bf16
    MFMA kernels forward and backward, csrc/dense_train.hip; V3D_DENSE_TRAIN=torch: torch modules under bf16 autocast) ->
    ProposalLoss -> backward -> all-reduce -> clip_grad_norm_(35) -> Adam   (reference train.py:58-70)."""
    from vision3d_amd import dist_util, synth
    from vision3d_amd.core import Preprocessor, ProposalTargetAssigner
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import ProposalLoss, Second
    import torch.distributed as dist
    rank, local, world = dist_util.env_world()
    torch.cuda.set_device(local % torch.cuda.device_count())
    pin_rank_to_gpu_numa(local, world)
    dist_util.init_from_env(BACKEND)
    # (only matters with V3D_DENSE_TRAIN=torch, where the dense RPN / heads train through MIOpen: let it search its solvers during
    # the warm-up -- 811 -> 868 frames/s on the same box against the default heuristic pick; V3D_TRAIN_BENCHMARK=0 switches it off)
    torch.backends.cudnn.benchmark = os.environ.get("V3D_TRAIN_BENCHMARK", "1") != "0"
    cfg = second_car_cfg()
    if args.points is None:
        args.points = 16384
    bs = args.batch if args.batch > 1 else 8
    native_env = os.environ.get("V3D_DENSE_TRAIN", "native") == "native"
    loss_fn = ProposalLoss(cfg)
    pre, assigner = Preprocessor(cfg, seed=0), ProposalTargetAssigner(cfg)
    fids = [rank * bs + i for i in range(bs)]
    clouds = [torch.from_numpy(synth.make_cloud(f, args.points)).cuda() for f in fids]
    targets = []
    for f in fids:
        gt = torch.from_numpy(synth.make_gt_boxes(f))
        targets.append(assigner(dict(boxes=gt, class_idx=torch.zeros(len(gt), dtype=torch.long),
                                     box_ignore=torch.zeros(len(gt), dtype=torch.bool))))
    tgt = {k: torch.stack([t[k] for t in targets]).cuda() for k in ("G_cls", "G_reg", "M_cls", "M_reg")}
    fused_loss = os.environ.get("V3D_FUSED_LOSS", "1") != "0"  # csrc/proposal_loss.hip (needs the native dense path's fused maps)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_arith(amp):
        """K timed optimiser steps of a fresh model in one arithmetic: amp = the step under bf16 autocast (dense half: bf16 storage,
        one MFMA term); not amp = the reference's fp32 script as written (dense half: split hi + lo storage, three MFMA terms --
        Second.dense_train_precision "bf16x3"; "torch" = MIOpen fp32)."""
        torch.manual_seed(0)
        model = Second(cfg).cuda().train()
        native_dense = native_env and (amp or model.dense_train_precision != "torch")
        if not args.no_channels_last and not native_dense:  # torch dense path only: MIOpen's bf16 igemm kernels are NHWC-native
            model.rpn = model.rpn.to(memory_format=torch.channels_last)
            model.head = model.head.to(memory_format=torch.channels_last)
            model.rpn.register_forward_pre_hook(lambda m, a: (a[0].contiguous(memory_format=torch.channels_last),))
        # fused = one kernel per parameter-group chunk instead of ~15 multi-tensor launches (0.23 ms of a 7.4 ms step); same update rule
        opt = torch.optim.Adam(model.parameters(), lr=0.01, betas=(0.9, 0.99), weight_decay=0.01,
                               fused=os.environ.get("V3D_FUSED_ADAM", "1") != "0")
        params = [p for p in model.parameters() if p.requires_grad]
        sparse_ids = {id(p) for p in model.cnn.parameters()}
        reducer = dist_util.TwoPhaseGradReducer([p for p in params if id(p) not in sparse_ids],
                                                [p for p in params if id(p) in sparse_ids], world)

        def step():
            item = pre(dict(points=clouds))
            item.update(tgt)
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                out = model(item)
                if not fused_loss:
                    out.pop("_head_maps", None)  # A/B: ProposalLoss through the torch expressions
                losses = loss_fn(out)
            if world > 1:  # the dense half's bucket is reduced while the native sparse backward runs
                for plan in model.cnn.__dict__.get("_train_plans", {}).values():
                    plan.pre_backward_hook = reducer.start_early
            losses["loss"].backward()
            reducer.finish()
            torch.nn.utils.clip_grad_norm_(params, max_norm=35)
            model.check_train_overflow()  # this step's capacity word (copied behind the forward): before the weights are touched
            opt.step()
            return losses["loss"].detach()

        for _ in range(args.warmup):
            loss = step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        elapsed = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=REDUCE_DEVICE)
        model.check_train_overflow()  # the last step's capacity word
        if amp:
            dense = ("bf16 dense RPN/head on hand-written MFMA kernels (csrc/dense_train.hip), fp32 accumulate; " if native_dense else
                     "bf16 autocast dense RPN/head (torch / MIOpen); ")
        else:
            dense = ("fp32-class dense RPN/head on hand-written kernels: split bf16 hi + lo storage, three MFMA terms per product, fp32 "
                     "accumulate (csrc/dense_train.hip split path + csrc/dense_conv.hip), no autocast -- the reference's train.py as "
                     "written; " if native_dense else "fp32 dense RPN/head (torch / MIOpen); ")
        return dict(elapsed=elapsed, loss=float(loss), model=model, native_dense=native_dense, fallbacks=int(model.torch_dense_fallbacks),
                    dtype=dense + "sparse convs bf16-split MFMA fwd/dX, fp32 MFMA dW")

    amp = args.train_arith == "bf16" and not args.no_amp
    main = run_arith(amp)
    elapsed, loss, model, native_dense = main["elapsed"], main["loss"], main["model"], main["native_dense"]
    second = None
    if not args.no_fast_mode:
        del model
        other = run_arith(not amp)
        second = dict(value=world * bs * args.steps / other["elapsed"], unit="frames/s", ms_per_step=1e3 * other["elapsed"] / args.steps,
                      dtype=other["dtype"], final_loss=other["loss"], torch_dense_fallbacks=other["fallbacks"],
                      note=("the same step as the reference's train.py:58-66 runs it: fp32, no autocast anywhere -- dense half on the native "
                            "fp32-class kernels, no MIOpen convolution in the step" if amp else
                            "the same step under torch.autocast(bfloat16) (BASELINE configs[2]): bf16-storage dense half"))
        model = other["model"]  # (any trained model of the architecture serves the CPU baseline's state_dict)
    seen = ranks_seen(world, args.gpus)
    roofline = cpu_baseline = None
    if rank == 0 and not args.no_roofline:
        roofline = train_roofline(bs) if amp else train_roofline_split(bs)
        if roofline and bs == 8:  # (the counter passes were taken on the 8-frame step)
            mc = mfma_busy_from_profiles("train", "dt_conv3_kernel" if amp else "conv2d_bf16x3_large_kernel<3, 9, 0>", roofline.get("avg_us"))
            roofline["mfma_busy_frac"], roofline["mfma_counters"] = (mc or {}).get("mfma_busy_frac"), mc
            if mc and amp and "dt_conv3_kernel" in "".join(json.load(open(os.path.join(REPO, "profiles", "pmc_mfma.json"))).get("train", {}).keys()):
                rec = [v for k, v in json.load(open(os.path.join(REPO, "profiles", "pmc_mfma.json")))["train"].items() if "dt_conv3_kernel" in k][0]
                roofline["traffic"] = (rec.get("fetch_bytes_x2") or 0.0) + (rec.get("write_bytes") or 0.0)
                roofline["traffic_source"] = "profiles/r06_pmc_mfma.txt (FETCH_SIZE x2 + WRITE_SIZE per launch)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_baseline = train_cpu_baseline(model, cfg, clouds[0], {k: v[:1] for k, v in tgt.items()}, args)
        except Exception as e:  # a reported extra
            cpu_baseline = dict(value=None, unit="frames/s", error=f"{type(e).__name__}: {str(e)[:200]}")
    if rank == 0:
        print(json.dumps(dict(
            metric="frames/sec SECOND train step, 16k-pt KITTI cloud", value=world * bs * args.steps / elapsed, unit="frames/s",
            n_gpus=world, n_ranks_seen=seen, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
            scaling="weak", vs_baseline=None,
            dtype=main["dtype"], data="synthetic", **({"fp32_script" if amp else "fast_mode": second} if second else {}),
            config=dict(workload="SECOND train step (BASELINE configs[2]): fwd + ProposalLoss + bwd + grad all-reduce + clip + Adam",
                        proposal_loss=("native fused pass (csrc/proposal_loss.hip)" if (fused_loss and native_dense) else "torch expressions"),
                        frames_per_gpu_per_step=bs, points_per_frame=args.points, parallelism=f"data-parallel x{world}, two-bucket all-reduce (dense bucket overlapped with the sparse backward)"),
            roofline=roofline, cpu_baseline=cpu_baseline, torch_dense_fallbacks=main["fallbacks"], final_loss=float(loss))))
    if world > 1:
        dist.destroy_process_group()


def train_roofline(bs, h=200, w=176):
    """The dominant kernel of the train step -- dt_conv3_kernel (csrc/dense_train.hip: the 3x3 128 -> 128 convolution on bf16 NHWC,
    12 launches per step: 6 forward + 6 data gradients) -- against the dense bf16 MFMA peak: algorithmic flops 2 * M * 128 * 128 * 9
    per launch over the average duration of 20 back-to-back launches between two HIP events on the launch stream."""
    from vision3d_amd import _lib as L
    lib = L.lib()
    x = torch.randn(bs, 128, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(128, 128, 3, 3, device="cuda") / 34.0
    img = torch.empty(int(lib.v3d_dense_train_weight_image_bytes(3)), dtype=torch.uint8, device="cuda")
    L.check(lib.v3d_dense_train_pack_weights(L.ptr(wt), 3, 0, L.ptr(img), L.stream_ptr()), "pack")
    y = torch.empty_like(x)
    stats = torch.empty((lib.v3d_dense_train_conv_tiles(bs, h, w), 2, 128), device="cuda")
    run = lambda: L.check(lib.v3d_dense_train_conv(L.ptr(x), L.ptr(img), bs, h, w, 3, L.ptr(y), L.ptr(stats), L.stream_ptr()), "conv")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 20
    fl = 2.0 * bs * h * w * 128 * 128 * 9
    in_bytes, out_bytes = 2.0 * bs * h * w * 128, 2.0 * bs * h * w * 128
    return dict(bound="mfma", kernel="dt_conv3_kernel", launches_per_step=12, flops_per_launch=fl, avg_us=t * 1e6,
                achieved=fl / t / 1e12, peak=2500.0, unit="TFLOP/s", frac=fl / t / 1e12 / 2500.0,
                algorithmic_bytes_per_launch=in_bytes + out_bytes + 2.0 * 9 * 128 * 128,
                hbm_view=dict(achieved_gbs=(in_bytes + out_bytes) / t / 1e9, frac=(in_bytes + out_bytes) / t / 1e9 / HBM_PEAK_GBS),
                traffic=None, traffic_note="PMC passes of this kernel: profiles/r03_a_pmc_dense_train.txt (2 x 58.8 MB fetched, 81.7 MB "
                                           "written per launch at bs = 8)",
                note="bf16 operands, fp32 accumulate, one MFMA term per product (the autocast contract); dense bf16 peak 2.5 PFLOP/s")


def train_roofline_split(bs, h=200, w=176):
    """The dominant kernel of the fp32-class train step -- the 3x3 128 -> 128 bf16x3 convolution of csrc/dense_conv.hip on split
    planes (12 launches per step: 6 forward + 6 data gradients) -- against the dense bf16 MFMA peak.  Algorithmic flops
    2 * M * 128 * 128 * 9 per launch (the three terms of the split product are the arithmetic's price, not useful work) over the
    average of 20 back-to-back launches between two HIP events on the launch stream."""
    from vision3d_amd import _lib as L
    from vision3d_amd.runtime import pack_conv_weight, split_planes_like, to_split_nhwc
    lib = L.lib()
    hi, lo = to_split_nhwc(torch.randn(bs, 128, h, w, device="cuda"))
    img = pack_conv_weight(torch.randn(128, 128, 3, 3, device="cuda") / 34.0, None, "bf16x3")
    y_hi, y_lo = split_planes_like(bs, h, w, 128, hi.device)
    run = lambda: L.check(lib.v3d_conv2d_nhwc_split(L.ptr(hi), L.ptr(lo), L.ptr(img), None, 0, bs, h, w, 128, 128, 3, L.ptr(y_hi), L.ptr(y_lo),
                                                    None, None, 0, None, None, None, None, None, 0, None, L.stream_ptr()), "conv")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 20
    fl = 2.0 * bs * h * w * 128 * 128 * 9
    io = 2 * 2 * 2.0 * bs * h * w * 128  # two planes in, two planes out, bf16
    return dict(bound="mfma", kernel="conv2d_bf16x3_large_kernel<3>", launches_per_step=12, flops_per_launch=fl, avg_us=t * 1e6,
                achieved=fl / t / 1e12, peak=2500.0, unit="TFLOP/s", frac=fl / t / 1e12 / 2500.0, issued_over_useful=3.0,
                algorithmic_bytes_per_launch=io + 2 * 2.0 * 9 * 128 * 128,
                hbm_view=dict(achieved_gbs=io / t / 1e9, frac=io / t / 1e9 / HBM_PEAK_GBS), traffic=None,
                note="useful flops (one product per weight x activation pair) against the dense bf16 peak; the kernel issues three MFMA "
                     "terms per product (hi*hi + hi*lo + lo*hi)")


def train_cpu_baseline(model, cfg, cloud, tgt1, args):
    """cpu_baseline of the train line: oracle/train_cpu.py (scalar C voxelizer + rulebooks, gather-GEMM-scatter sparse convolutions
    and the dense half in torch CPU ops, autograd, clip, Adam) on ONE thread, bounded sample: steps of ONE frame (bs = 1)."""
    from oracle import train_cpu
    torch.set_num_threads(1)
    sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    tg = {k: v.detach().cpu().numpy() for k, v in tgt1.items()}
    cl = cloud.detach().cpu().numpy()
    kw = dict(lam=float(cfg.TRAIN.LAMBDA), max_pts=cfg.MAX_OCCUPANCY, max_voxels=cfg.MAX_VOXELS)
    _, _, params, opt = train_cpu.train_step(sd, [cl], tg, list(cfg.VOXEL_SIZE), list(cfg.GRID_BOUNDS), **kw)  # warm-up (library start-up)
    n, t = 0, 0.0
    while n < 4 and t < 15.0:
        _, dt, params, opt = train_cpu.train_step(params, [cl], tg, list(cfg.VOXEL_SIZE), list(cfg.GRID_BOUNDS), optimizer=opt, **kw)
        n, t = n + 1, t + dt
    n_phys, n_threads = physical_cores()
    return dict(value=n / t, unit="frames/s", cores=1, kind="port", host_cores=n_phys, host_threads=n_threads,
                sample=f"{n} train step(s) of ONE {args.points}-pt frame (bs = 1) on 1 thread, oracle/train_cpu.py, {t:.1f} s")


def pvrcnn_main(args):
    """configs[3]: PV-RCNN stage 2 on SECOND proposals -- FPS keypoints (16 384 -> 2 048), 5-level voxel-set abstraction
    (ball query + group + shared MLP + max), BEV bilinear gather, RoI-grid pooling of 100 proposals, refinement MLP.
    Stage-1 outputs (sparse feature volumes, BEV map, proposals) are resident before the timed region.  `value` = throughput
    with --pipeline frames in flight (default 4: independent frames on separate streams, one host thread, no host
    synchronisation inside a step); `single_frame_ms` = the same step one frame at a time."""
    from vision3d_amd import dist_util, synth
    from vision3d_amd.core import Preprocessor
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.detector import PV_RCNN
    import torch.distributed as dist
    rank, local, world = dist_util.env_world()
    torch.cuda.set_device(local % torch.cuda.device_count())
    pin_rank_to_gpu_numa(local, world)
    dist_util.init_from_env(BACKEND)
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = PV_RCNN(cfg).cuda().eval()
    bs = args.batch
    args.n_ranks_seen = ranks_seen(world, args.gpus)
    if args.end_to_end:
        return pvrcnn_end_to_end(args, model, cfg, rank, world)
    depth = args.pipeline if args.pipeline >= 1 else 4  # frames in flight (independent frames on separate streams; 1 = one at a time)
    n_prop = 100
    slots = []
    with torch.no_grad():
        for sl in range(depth):  # every slot works on its own frame(s): stage-1 outputs resident before the timed region
            clouds = [synth.make_cloud((rank * depth + sl) * bs + i, args.points or 16384) for i in range(bs)]
            item = model.proposal(Preprocessor(cfg, seed=0)(dict(points=clouds)))
            gts = [synth.make_gt_boxes((rank * depth + sl) * bs + i) for i in range(bs)]
            props = torch.from_numpy(np.stack([np.resize(g, (n_prop, 7)) for g in gts])).cuda()
            slots.append((item, props, torch.cuda.Stream()))

        def step(sl=0):
            item, props, _ = slots[sl]
            item["keypoints"] = model.sample_keypoints(item["points"])
            pf = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
            pooled = model.roi_grid_pool(props, item["keypoints"], pf)
            return model.refinement_layer(None, pooled, props)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # one frame at a time on the current stream (latency), then `depth` frames in flight: FPS is ONE workgroup's dependent
        # chain (2.5 ms on one CU), so the set abstraction / RoI pooling of other frames runs beside it on the rest of the chip
        for _ in range(max(args.warmup, 2)):
            out = step(0)
        fence()
        t0 = time.perf_counter()
        for _ in range(max(args.steps // 2, 1)):
            out = step(0)
        torch.cuda.synchronize()
        single_ms = 1e3 * (time.perf_counter() - t0) / max(args.steps // 2, 1)
        for sl in range(depth):
            slots[sl][2].wait_stream(torch.cuda.current_stream())
        graphs = None
        if os.environ.get("V3D_PVRCNN_GRAPH", "1") != "0":  # one captured HIP graph per slot: ~130 eager launches -> one replay
            try:
                graphs = []
                for sl in range(depth):
                    with torch.cuda.stream(slots[sl][2]):
                        step(sl)  # warm-up on the slot's stream (allocator pools, lazily built state)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=slots[sl][2]):
                        step(sl)
                    graphs.append(g)
            except Exception as e:  # capture is an optimisation of the launch path only
                print(f"bench: PV-RCNN graph capture unavailable ({type(e).__name__}: {str(e)[:120]}); eager launches", file=sys.stderr)
                graphs = None
                torch.cuda.synchronize()

        # Which streams: two HIP streams overlap only if their hardware queues sit on different command-processor pipes (the line read
        # 505 or 915 frames/s from one fresh process to the next with streams taken in creation order), so the replay streams of the
        # slots are picked by measurement like the forward pipeline's (detector/graph.py choose_streams): every pair out of 8
        # candidates, then greedily deeper while it pays.  A captured graph replays on whatever stream is current.
        run_streams = [sl[2] for sl in slots]
        tuned = None
        if graphs is not None and depth >= 2 and args.pipeline < 1:
            from vision3d_amd.detector.graph import choose_streams
            cands = [torch.cuda.Stream() for _ in range(8)]

            def time_of(ids, frames=12):
                best = None
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for f in range(frames):
                        with torch.cuda.stream(cands[ids[f % len(ids)]]):
                            graphs[f % len(ids)].replay()
                    torch.cuda.synchronize()
                    t = (time.perf_counter() - t0) / frames
                    best = t if best is None else min(best, t)
                return best
            chosen, log = choose_streams(time_of, len(cands), depth)
            run_streams = [cands[c] for c in chosen]
            tuned = dict(depth=len(chosen), us_per_frame={k: round(v * 1e6, 1) for k, v in log.items()})
        n_run = len(run_streams)

        def submit(i):
            sl = i % n_run
            with torch.cuda.stream(run_streams[sl]):
                if graphs is not None:
                    graphs[sl].replay()
                else:
                    step(sl)

        for i in range(args.warmup):
            submit(i)
        fence()
        t0 = time.perf_counter()
        enq = 0.0
        for i in range(args.steps):
            e0 = time.perf_counter()
            submit(i)
            enq += time.perf_counter() - e0
        fence()
    elapsed = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=REDUCE_DEVICE)
    # ---- the same frames with the keypoint samplings of the frames in flight BATCHED into one launch (round 6).  Farthest-point sampling
    # is 2 048 dependent steps on ONE compute unit per cloud (2.48 ms of a 3.65 ms frame) and takes a batch: one launch samples the
    # D frames of the next round, a workgroup each, on a stream of its own, while the rest of stage 2 of the CURRENT round's D frames
    # (one captured graph per frame, ~1.1 ms) runs on the other streams; keypoint buffers are double-buffered, every dependency is an
    # event.  Same per-frame results (the sampling of a cloud does not depend on its batch).
    batched = None
    if graphs is not None and args.pipeline < 1:
        try:
            with torch.no_grad():
                D = 16
                while len(slots) < D:
                    sl = len(slots)
                    clouds = [synth.make_cloud((rank * D + sl) * bs + i, args.points or 16384) for i in range(bs)]
                    item = model.proposal(Preprocessor(cfg, seed=0)(dict(points=clouds)))
                    gts = [synth.make_gt_boxes((rank * D + sl) * bs + i) for i in range(bs)]
                    props = torch.from_numpy(np.stack([np.resize(g, (n_prop, 7)) for g in gts])).cuda()
                    slots.append((item, props, torch.cuda.Stream()))
                torch.cuda.synchronize()
                pts_all = torch.cat([slots[sl][0]["points"] for sl in range(D)], dim=0).contiguous()  # (D * bs, N, C)
                n_kp = cfg.NUM_KEYPOINTS
                kp = [torch.zeros((D * bs, n_kp, 3), dtype=torch.float32, device=pts_all.device) for _ in range(2)]

                def rest(sl, buf):
                    item, props, _ = slots[sl]
                    item["keypoints"] = kp[buf][sl * bs:(sl + 1) * bs]
                    pf = model.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
                    pooled = model.roi_grid_pool(props, item["keypoints"], pf)
                    return model.refinement_layer(None, pooled, props)
                cap = torch.cuda.Stream()
                cap.wait_stream(torch.cuda.current_stream())
                gA, gB = [], [[None, None] for _ in range(D)]
                with torch.cuda.stream(cap):
                    for buf in range(2):
                        kp[buf].copy_(model.sample_keypoints(pts_all))
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=cap):
                            kp[buf].copy_(model.sample_keypoints(pts_all))
                        gA.append(g)
                        for sl in range(D):
                            rest(sl, buf)
                            torch.cuda.synchronize()
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, stream=cap):
                                rest(sl, buf)
                            gB[sl][buf] = g
                torch.cuda.synchronize()
                eA = [torch.cuda.Event() for _ in range(2)]
                eB = [[torch.cuda.Event() for _ in range(D)] for _ in range(2)]

                def rounds(n_rounds, s_a, s_r):
                    for r in range(n_rounds):
                        buf = r % 2
                        if r >= 2:
                            for e in eB[buf]:
                                s_a.wait_event(e)  # the round that last read this keypoint buffer
                        with torch.cuda.stream(s_a):
                            gA[buf].replay()
                            eA[buf].record()
                        for sl in range(D):
                            st = s_r[sl % len(s_r)]
                            st.wait_event(eA[buf])
                            with torch.cuda.stream(st):
                                gB[sl][buf].replay()
                                eB[buf][sl].record()
                from vision3d_amd.detector.graph import choose_streams
                cands = [torch.cuda.Stream() for _ in range(8)]

                def time_of(ids):
                    best = None
                    for _ in range(2):
                        torch.cuda.synchronize()
                        c0 = time.perf_counter()
                        rounds(3, cands[ids[0]], [cands[j] for j in ids[1:]])
                        torch.cuda.synchronize()
                        t = (time.perf_counter() - c0) / (3 * D)
                        best = t if best is None else min(best, t)
                    return best
                chosen, log = choose_streams(time_of, len(cands), 4)
                s_a, s_r = cands[chosen[0]], [cands[j] for j in chosen[1:]]
                n_rounds = max(2, -(-args.steps // D))
                rounds(2, s_a, s_r)
                fence()
                c0 = time.perf_counter()
                rounds(n_rounds, s_a, s_r)
                fence()
                el_b = dist_util.max_over_ranks(time.perf_counter() - c0, world, device=REDUCE_DEVICE)
                batched = dict(value=world * bs * n_rounds * D / el_b, ms_per_step=1e3 * el_b / (n_rounds * D), frames_per_round=D,
                               rounds=n_rounds, streams=len(chosen), pipeline_tuning={k: round(v * 1e6, 1) for k, v in log.items()})
        except Exception as e:  # (the per-frame form above stays the line)
            print(f"bench: batched keypoint sampling unavailable ({type(e).__name__}: {str(e)[:160]})", file=sys.stderr)
            batched = None
    roofline = cpu_baseline = None
    if rank == 0 and not args.no_roofline:
        roofline = fps_roofline(model, slots[0][0]["points"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_baseline = pvrcnn_cpu_baseline(model, cfg, slots[0][0], slots[0][1], args)
        except Exception as e:  # a reported extra
            cpu_baseline = dict(value=None, unit="frames/s", error=f"{type(e).__name__}: {str(e)[:200]}")
    if rank == 0:
        print(json.dumps(dict(
            metric="frames/sec PV-RCNN stage 2, 16k-pt KITTI cloud",
            value=batched["value"] if batched else world * bs * args.steps / elapsed, unit="frames/s",
            n_gpus=world, n_ranks_seen=args.n_ranks_seen, steps=(batched["rounds"] * batched["frames_per_round"] if batched else args.steps), warmup=args.warmup,
            ms_per_step=batched["ms_per_step"] if batched else 1e3 * elapsed / args.steps, higher_is_better=True,
            batched_sampling=batched,
            sampling_per_frame=dict(value=world * bs * args.steps / elapsed, ms_per_step=1e3 * elapsed / args.steps, frames_in_flight=n_run,
                                    note="every frame's graph holds its own keypoint sampling (rounds 2-5's `value`)"),
            scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
            config=dict(workload="PV-RCNN stage 2 (BASELINE configs[3]): FPS 2048 keypoints + 5-level VSA + BEV gather + "
                                 "RoI-grid pool (100 proposals) + refinement MLP", frames_per_gpu_per_step=bs,
                        points_per_frame=args.points or 16384, parallelism=f"frame-parallel replicas x{world}",
                        frames_in_flight=(batched["frames_per_round"] if batched else n_run), pipeline_tuning=tuned,
                        path=(("keypoint samplings of the 16 frames of a round batched into ONE launch (a workgroup per cloud) on a stream of "
                               "its own, the rest of stage 2 one captured HIP graph per frame on the other streams, keypoint buffers double-"
                               "buffered, every dependency an event") if batched else
                              ("one captured HIP graph per frame in flight" if graphs is not None else "eager launches") +
                              ", one host thread, one stream per frame in flight")),
            single_frame_ms=single_ms, frames_per_s_one_at_a_time=bs * 1e3 / single_ms,
            host_enqueue_ms_per_step=1e3 * enq / args.steps, roofline=roofline, cpu_baseline=cpu_baseline)))
    if world > 1:
        dist.destroy_process_group()


def plumbing_main(args):
    """configs[0]: voxelize (+ fused VFE mean) + points_in_boxes on one 16 384-point synthetic KITTI cloud -- the reference's
    CPU-runnable plumbing case (core/preprocess.py:26-33, core/geometry.py:27-65), here on the device beside the CPU port.
    A step = one frame: the voxelizer's launches and the points-in-boxes launch, replayed as one captured HIP graph per distinct
    frame of the stream (inputs resident in HBM, nothing read back inside the timed region).  Roofline: HBM, algorithmic bytes
    16 N + 36 M (voxelizer: points read once; per voxel coords + occupancy + mean written, max_pts point slots not counted: the
    detector consumes the mean) and 12 N + 28 n + N n (points-in-boxes: xyz read, boxes read, mask written), SURVEY.md 8(d)."""
    from vision3d_amd import dist_util, synth
    from vision3d_amd.core.config import second_car_cfg
    from vision3d_amd.core.geometry import points_in_boxes_mask
    from vision3d_amd.spconv.utils import voxelize_batch
    import torch.distributed as dist
    rank, local, world = dist_util.env_world()
    torch.cuda.set_device(local % torch.cuda.device_count())
    pin_rank_to_gpu_numa(local, world)
    dist_util.init_from_env(BACKEND)
    n_ranks_seen = ranks_seen(world, args.gpus)
    cfg = second_car_cfg()
    n_pts = args.points or 16384
    n_stream = max(1, args.stream)
    seeds = [j * world + rank for j in range(n_stream)]
    clouds_np = [synth.make_cloud(s, n_pts) for s in seeds]
    boxes_np = [synth.make_gt_boxes(s) for s in seeds]
    clouds = [torch.from_numpy(c).cuda() for c in clouds_np]
    boxes = [torch.from_numpy(b).cuda() for b in boxes_np]

    def frame(j):
        _, coords, occ, mean, n_vox = voxelize_batch(clouds[j], [0, n_pts], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY,
                                                     cfg.MAX_VOXELS, want_voxels=False)
        return coords, occ, mean, n_vox, points_in_boxes_mask(clouds[j], boxes[j], True)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graphs, outs = [], []
    side = torch.cuda.Stream()
    with torch.no_grad():
        for j in range(n_stream):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                frame(j)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs.append(frame(j))
            graphs.append(g)
    for i in range(args.warmup):
        graphs[i % n_stream].replay()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        graphs[i % n_stream].replay()
    fence()
    elapsed = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=REDUCE_DEVICE)
    m_vox = int(outs[0][3].item())
    n_box = int(boxes[0].shape[0])
    roofline = cpu_baseline = None
    if rank == 0 and not args.no_roofline:
        def graph_us(fn, rep=20):
            g = torch.cuda.CUDAGraph()
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(rep):
                    fn()
            ts = []
            for trial in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                if trial:
                    ts.append(e0.elapsed_time(e1) * 1e3 / rep)
            return float(np.mean(ts))
        with torch.no_grad():
            t_vox = graph_us(lambda: voxelize_batch(clouds[0], [0, n_pts], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY,
                                                    cfg.MAX_VOXELS, want_voxels=False))
            t_pib = graph_us(lambda: points_in_boxes_mask(clouds[0], boxes[0], True))
        b_vox, b_pib = 16.0 * n_pts + 36.0 * m_vox, 12.0 * n_pts + 28.0 * n_box + float(n_pts) * n_box
        roofline = dict(bound="hbm", kernel="vox_insert + vox_emit (voxelizer + VFE mean, 2 launches; 1 fill)", achieved=b_vox / (t_vox * 1e-6) / 1e9,
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=b_vox / (t_vox * 1e-6) / 1e9 / HBM_PEAK_GBS, bytes_per_launch=b_vox,
                        avg_us=t_vox, traffic=None,
                        points_in_boxes=dict(kernel="points_in_boxes_kernel", avg_us=t_pib, bytes_per_launch=b_pib,
                                             achieved=b_pib / (t_pib * 1e-6) / 1e9, frac=b_pib / (t_pib * 1e-6) / 1e9 / HBM_PEAK_GBS),
                        note="0.7 MB of algorithmic traffic per frame against launches of ~10 us: the stage is bound by launch latency "
                             "and the dependent chain insert -> count/scan -> emit, not by HBM (SURVEY.md F9); durations are per "
                             "frame (all launches of the stage), back to back inside one HIP graph")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle as orc  # the CPU port: baseline only (never on the product path)
            orc.build()
            t0c, nfr = time.perf_counter(), 0
            while time.perf_counter() - t0c < 10.0:
                j = nfr % n_stream
                vox, co, num = orc.voxelize(clouds_np[j], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS)
                orc.vfe_mean(vox, num)
                orc.points_in_boxes(clouds_np[j], boxes_np[j], True)
                nfr += 1
            dtc = time.perf_counter() - t0c
            cpu_baseline = dict(value=nfr / dtc, unit="frames/s", cores=1, kind="port",
                                sample=f"{nfr} frames of the same stream in {dtc:.1f} s: oracle/v3d_oracle.c voxelizer + VFE mean + "
                                       "points-in-boxes, one thread")
        except Exception as e:  # a reported extra
            cpu_baseline = dict(value=None, unit="frames/s", error=f"{type(e).__name__}: {str(e)[:200]}")
    if rank == 0:
        print(json.dumps(dict(
            metric="frames/sec voxelize + points_in_boxes, 16k-pt KITTI cloud", value=world * args.steps / elapsed, unit="frames/s",
            n_gpus=world, n_ranks_seen=n_ranks_seen, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32 / int32 (bit-exact voxel indices and masks)", data="synthetic",
            config=dict(workload="voxelize (+ VFE mean) + points_in_boxes on one 16384-pt synthetic KITTI cloud (BASELINE configs[0])",
                        points_per_frame=n_pts, voxels_per_frame=m_vox, boxes_per_frame=n_box, distinct_frames_in_timed_loop=n_stream,
                        parallelism=f"frame-parallel replicas x{world}", path="one captured HIP graph per frame, one at a time"),
            roofline=roofline, cpu_baseline=cpu_baseline)))
    if world > 1:
        dist.destroy_process_group()


def fps_roofline(model, points):
    """The dominant kernel of PV-RCNN stage 2: farthest-point sampling 16 384 -> 2 048 (fps_slab_kernel, csrc/pointops.hip), ONE
    workgroup per frame.  HBM view per the contract: compulsory bytes N * 12 + K * 4 (points read once, indices written; points and
    running distances stay in registers) over the measured duration; the naive K * N * 16 B of a re-reading loop is the diagnostic.
    What binds it is neither: K dependent steps of VALU work on one CU (`valu_floor_us`)."""
    from vision3d_amd.pointnet2 import pointnet2_utils as pn2
    xyz = points[..., :3].contiguous()
    b, n = xyz.shape[:2]
    k = model.cfg.NUM_KEYPOINTS
    for _ in range(2):
        pn2.furthest_point_sample(xyz, k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pn2.furthest_point_sample(xyz, k)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    alg = b * (n * 12.0 + k * 4.0)
    # per step every point: 3 subtractions, 3 multiply-adds, a min, a compare / select = 8 lane operations; one CU issues
    # 4 SIMDs x 16 lanes per clock (packed fp32 halves it): K * N * 8 / 64 clocks at 2.4 GHz
    valu_floor = k * n * 8.0 / 64.0 / 2.4e9
    return dict(bound="hbm", kernel="fps_slab_kernel<64>", launches_per_frame=1, bytes_per_launch=alg, avg_us=t * 1e6,
                achieved=alg / t / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=alg / t / 1e9 / HBM_PEAK_GBS, traffic=None,
                naive_bytes_per_launch=b * k * n * 16.0, us_per_step=t * 1e6 / k, valu_floor_us=valu_floor * 1e6,
                frac_of_valu_floor=valu_floor / t,
                note="latency chain of K = 2 048 dependent steps on ONE CU (a second CU would need a cross-CU exchange per step, "
                     "~2 us); the HBM fraction of a kernel that reads 196 KB once is not a meaningful target, the per-step cost "
                     "against the one-CU VALU floor is")


def pvrcnn_cpu_baseline(model, cfg, item, props, args):
    """cpu_baseline of the stage-2 line: oracle/pvrcnn_cpu.py (scalar C FPS / ball query / group + the model's own MLPs as torch CPU
    modules) on ONE thread, one frame."""
    import copy
    from oracle import pvrcnn_cpu
    torch.set_num_threads(1)
    cpu = copy.deepcopy(model).cpu().eval()
    pts = item["points"][:1].detach().cpu().numpy()
    feats = [(x[:1].detach().cpu().numpy(), f[:1].detach().cpu().numpy()) for x, f in item["_cnn_features"]]
    bev = item["_bev_map"][:1].detach().cpu().numpy()
    pr = props[:1].detach().cpu().numpy()
    samples = torch.rand((1, pr.shape[1], cfg.GRIDPOOL.NUM_GRIDPOINTS, 3), generator=torch.Generator().manual_seed(0)).numpy()
    n, t = 0, 0.0
    while n < 3 and t < 15.0:
        _, _, dt = pvrcnn_cpu.stage2(cpu, pts, feats, bev, pr, samples, cfg.NUM_KEYPOINTS)
        n, t = n + 1, t + dt
    n_phys, n_threads = physical_cores()
    return dict(value=n / t, unit="frames/s", cores=1, kind="port", host_cores=n_phys, host_threads=n_threads,
                sample=f"{n} frame(s) of the same stage-2 workload (FPS 2048 + 5-level VSA + BEV gather + RoI-grid pool of "
                       f"{pr.shape[1]} proposals + refinement), oracle/pvrcnn_cpu.py on 1 thread, {t:.1f} s")


def pvrcnn_end_to_end(args, model, cfg, rank, world):
    """PV_RCNN.inference(item) from raw points, one frame at a time: device voxelizer -> sparse CNN (4 levels + BEV) -> stage-1
    head (MFMA 1x1) -> FPS keypoints + VSA + BEV gather -> top-k proposals -> RoI-grid pool -> refinement -> rotated NMS.
    (Upstream's PV_RCNN.forward raises: this is the repository's wiring of the reference's pieces, detector/model.py.)"""
    from vision3d_amd import dist_util, synth
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    import torch.distributed as dist
    pre = Preprocessor(cfg, seed=0)
    anchors = AnchorGenerator(cfg).anchors.cuda()
    bs = args.batch
    frames = [[torch.from_numpy(synth.make_cloud((j * world + rank) * bs + i, args.points or 16384)).cuda() for i in range(bs)]
              for j in range(max(1, args.stream))]

    def make_item(i):
        return pre(dict(points=frames[i % len(frames)], anchors=anchors))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    AHEAD = 1  # frames whose keypoint sampling is in flight ahead of the frame being issued (a side stream / compute unit each).  Two ahead
               # measured in round 6: 358 vs 375 frames/s -- the main stream's ~190 eager launches per frame (2.7 ms of host time) bound
               # the line, not the 2.48 ms sampling

    def run(n, prefetch):
        """n frames; prefetch: frames i + 1 .. i + AHEAD are preprocessed and their keypoint samplings (one compute unit and 2.5 ms
        each, on side streams of their own) started before frame i's inference is issued, so a sampling overlaps whole frames of
        other work instead of its own stage 1 only."""
        out = None
        with torch.no_grad():
            ahead = [model.prefetch_keypoints(make_item(j)) for j in range(min(AHEAD, n))] if prefetch else []
            for i in range(n):
                if prefetch:
                    item = ahead.pop(0)
                    if i + AHEAD < n:
                        ahead.append(model.prefetch_keypoints(make_item(i + AHEAD)))
                else:
                    item = make_item(i)
                out = model.inference(item)
        return out

    def run_in_flight(n):
        """Two frames in flight from this one host thread (PV_RCNN.inference_begin / _end / _collect): stage 1 of frame i + 1 is queued
        before frame i's row counts are read, frame i's result is read after frame i + 1's stage 2 is queued; keypoint samplings
        for AHEAD_IN_FLIGHT frames at a time in ONE launch (a workgroup per cloud) on a side stream."""
        out = None
        with torch.no_grad():
            items = model.prefetch_keypoints_many([make_item(j) for j in range(min(AHEAD_IN_FLIGHT, n))])
            nxt_j = len(items)
            st = model.inference_begin(items.pop(0), 0)
            prev = None
            for i in range(n):
                if len(items) < AHEAD_IN_FLIGHT // 2 + 1 and nxt_j < n:  # the samplings of the next AHEAD_IN_FLIGHT frames in ONE launch
                    batch = [make_item(j) for j in range(nxt_j, min(nxt_j + AHEAD_IN_FLIGHT, n))]
                    items += model.prefetch_keypoints_many(batch)
                    nxt_j += len(batch)
                nxt = model.inference_begin(items.pop(0), (i + 1) % 2) if i + 1 < n else None
                h = model.inference_end(st)
                if prev is not None:
                    out = model.inference_collect(prev)
                prev, st = h, nxt
            out = model.inference_collect(prev)
        return out

    AHEAD_IN_FLIGHT = 4
    timings = {}
    for prefetch in (False, True, "in_flight"):
        fn = (lambda n: run_in_flight(n)) if prefetch == "in_flight" else (lambda n, p=prefetch: run(n, p))
        fn(max(args.warmup, 2))
        fence()
        t0 = time.perf_counter()
        out = fn(args.steps)
        fence()
        timings[prefetch] = dist_util.max_over_ranks(time.perf_counter() - t0, world, device=REDUCE_DEVICE)
    elapsed = timings["in_flight"]
    if rank == 0:
        print(json.dumps(dict(
            metric="frames/sec PV-RCNN inference end to end, 16k-pt KITTI cloud", value=world * bs * args.steps / elapsed,
            unit="frames/s", n_gpus=world, n_ranks_seen=args.n_ranks_seen, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32 (set abstraction) / f16s split, fp32-class (sparse CNN, head)",
            data="synthetic",
            one_frame_at_a_time=dict(value=world * bs * args.steps / timings[False], ms_per_step=1e3 * timings[False] / args.steps,
                                     note="no prefetch: the keypoint sampling of a frame overlaps that frame's stage 1 only"),
            prefetch_only=dict(value=world * bs * args.steps / timings[True], ms_per_step=1e3 * timings[True] / args.steps,
                               note="PV_RCNN.inference frame by frame, the next frame's keypoint sampling started before it (rounds 4-5's `value`)"),
            config=dict(workload="PV_RCNN inference from raw points (stage 1 + BASELINE configs[3] stage 2 + refinement NMS), eager "
                                 "launches from one host thread, two frames in flight (PV_RCNN.inference_begin / _end / _collect: stage 1 "
                                 "of frame i + 1 is queued before frame i's row counts are read, every wait is on the frame's own event); "
                                 "keypoint samplings (farthest-point sampling: 2 048 dependent steps on one compute unit per cloud) of the "
                                 "next frames batched into one launch on a side stream (PV_RCNN.prefetch_keypoints_many)",
                        frames_in_flight=2, keypoint_samplings_ahead=AHEAD_IN_FLIGHT,
                        frames_sampled_ahead=AHEAD,
                        frames_per_gpu_per_step=bs,
                        points_per_frame=args.points or 16384, parallelism=f"frame-parallel replicas x{world}"),
            n_detections=int(out[0].shape[0]), roofline=None, cpu_baseline=None)))
    if world > 1:
        dist.destroy_process_group()


def _cpu_frame(job):
    """One frame of the CPU restatement (oracle/: scalar C sparse path + torch CPU dense path) on ONE thread; returns seconds."""
    sd, seed, points, cfgd, anchors, workload = job
    import torch as _t
    _t.set_num_threads(1)
    from oracle import second_cpu
    from vision3d_amd import synth as _s
    cloud = _s.make_waymo_cloud(seed, points) if workload == "waymo" else _s.make_cloud(seed, points)
    c0 = time.perf_counter()
    ref = second_cpu.second_forward(sd, [cloud], cfgd["VOXEL_SIZE"], cfgd["GRID_BOUNDS"], cfgd["MAX_OCCUPANCY"], cfgd["MAX_VOXELS"])
    second_cpu.proposals(ref["cls"], ref["reg"], anchors, 1, 2, 7, cfgd["TOPK"], cfgd["THRESH"])
    return time.perf_counter() - c0


def physical_cores():
    """(physical cores available to this process, hardware threads available): distinct (package, core id) pairs of the CPUs in the
    affinity mask, from /proc/cpuinfo; falls back to the thread count."""
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    try:
        cores, cur = {}, {}
        for ln in list(open("/proc/cpuinfo")) + [""]:
            if ":" in ln:
                k, v = ln.split(":", 1)
                cur[k.strip()] = v.strip()
            elif cur:
                cores[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                cur = {}
        phys = len({cores[c] for c in avail if c in cores})
        return max(1, phys), len(avail)
    except (OSError, KeyError, ValueError):
        return len(avail), len(avail)


def run_cpu_baseline(model, cfg, anchors, make, args, workload="kitti"):
    """cpu_baseline object: the CPU restatement timed on the host of THIS box -- one thread (value) and ALL PHYSICAL CORES (one
    frame per single-thread worker process: frames are independent, so that is how a CPU deployment would use its cores)."""
    import multiprocessing as mp
    cpu_model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n_phys, n_threads = physical_cores()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cfgd = dict(VOXEL_SIZE=list(cfg.VOXEL_SIZE), GRID_BOUNDS=list(cfg.GRID_BOUNDS), MAX_OCCUPANCY=cfg.MAX_OCCUPANCY,
                MAX_VOXELS=cfg.MAX_VOXELS, TOPK=cfg.PROPOSAL.TOPK, THRESH=[a["score_thresh"] for a in cfg.ANCHORS])
    anc = anchors.cpu().numpy()
    n_frames, t_cpu = 0, 0.0
    while n_frames < args.cpu_frames and t_cpu < (20.0 if workload == "kitti" else 1.0):
        t_cpu += _cpu_frame((sd, 100 + n_frames, args.points, cfgd, anc, workload))
        n_frames += 1
    out = dict(value=n_frames / t_cpu, unit="frames/s", cores=1, kind="port", cpu_model=cpu_model, host_cores=n_phys,
               host_threads=n_threads,
               sample=f"{n_frames} frame(s) of the same {args.points}-pt workload, oracle/ (scalar C sparse path + torch CPU dense "
                      f"path, 1 thread), {t_cpu:.1f} s")
    workers = n_phys
    per_frame = t_cpu / n_frames
    if per_frame > 30.0:  # bounded sample: a frame that takes this long is not repeated on every core
        out["all_cores"] = dict(value=None, note=f"skipped: one frame takes {per_frame:.0f} s on one thread")
        return out
    try:
        ctx = mp.get_context("spawn")
        jobs = [(sd, 200 + i, args.points, cfgd, anc, workload) for i in range(workers)]
        w0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            per = pool.map(_cpu_frame, jobs, chunksize=1)
        wall = time.perf_counter() - w0
        out["all_cores"] = dict(value=workers / max(per), unit="frames/s", cores=workers,
                                sample=f"{workers} frames in {workers} single-thread worker processes (= all physical cores), "
                                       f"slowest frame {max(per):.2f} s, fastest {min(per):.2f} s (wall incl. process start {wall:.1f} s)")
    except Exception as e:  # the 1-thread figure stands on its own
        out["all_cores"] = dict(value=None, error=str(e)[:200])
    return out


def _arm_watchdog(limit_s):
    """A run that stops making progress must end, not sit on the GPU box: a daemon thread dumps every Python stack and exits with
    status 124 when the process is older than `limit_s` seconds (0 = off).  The default (1 500 s) is several times the longest mode."""
    if limit_s <= 0:
        return
    import faulthandler
    import threading

    def bark():
        sys.stderr.write(f"bench.py: no result after {limit_s} s -- dumping stacks and exiting\n")
        faulthandler.dump_traceback(all_threads=True)
        sys.stderr.flush()
        os._exit(124)
    t = threading.Timer(limit_s, bark)
    t.daemon = True
    t.start()


# ---- the other BASELINE configurations on the driver's clock: compact sub-lines of the default run --------------------------------
EXTRA_RUNS = [  # (key, what, argv of a short run of that mode)
    ("pvrcnn_stage2", "BASELINE configs[3]: PV-RCNN stage 2 on SECOND proposals (keypoint samplings of a round's 16 frames in one launch; per-frame form beside it)",
     ["--mode", "pvrcnn", "--windows", "5", "--steps", "64", "--warmup", "5"]),
    ("waymo", "BASELINE configs[4]: SECOND forward, 180 k-pt Waymo-range sweep, frames in flight as the headline (streams and depth by measurement)",
     ["--workload", "waymo", "--windows", "5", "--steps", "60", "--warmup", "10", "--single-frames", "40", "--stream", "4"]),
    ("kitti_bs8", "SECOND forward, batch of 8 KITTI clouds per step (65-110 k rows per sparse stage: the large-layer kernels), batches in flight as the headline",
     ["--batch", "8", "--windows", "5", "--steps", "30", "--warmup", "5", "--single-frames", "20", "--stream", "2"]),
    ("plumbing", "BASELINE configs[0]: voxelize + points_in_boxes, one 16 k-pt cloud", ["--mode", "plumbing", "--windows", "5", "--steps", "100"]),
    ("train", "BASELINE configs[2]: SECOND train step bf16, 8 frames per GPU", ["--mode", "train", "--steps", "6", "--warmup", "2"]),
]


def run_extras():
    """-> {key: compact line} -- every entry a short run of bench.py's own mode in a FRESH process (same code path as the full line of
    that mode, fewer windows, no CPU baseline, no bf16x3 / H2D side lines); never raises.  A fresh process because the in-process
    form measured the process's history, not the configuration: which hardware queues a new HIP stream lands on depends on the
    streams created before it (the Waymo-range line read 1 168 frames/s as the first extra and 972 behind the PV-RCNN one), and the
    host-bound PV-RCNN line halves behind a run that left the interpreter a large heap."""
    import subprocess
    out = {}
    for key, what, argv in EXTRA_RUNS:
        t0 = time.perf_counter()
        try:
            cmd = [sys.executable, os.path.abspath(__file__)] + argv + ["--no-cpu-baseline", "--no-fast-mode", "--no-h2d", "--no-extra", "--watchdog", "240"]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
            if res.returncode != 0 or not lines:
                raise RuntimeError(f"rc {res.returncode}: {res.stderr.strip()[-200:]}")
            d = json.loads(lines[-1])
            r = d.get("roofline") or {}
            out[key] = dict(what=what, metric=d.get("metric"), value=d.get("value"), unit=d.get("unit"), ms_per_step=d.get("ms_per_step"),
                            single_frame_ms=d.get("single_frame_ms"), steps=d.get("steps"), windows=d.get("windows"),
                            workload=(d.get("config") or {}).get("workload"),
                            roofline=dict(kernel=r.get("kernel"), bound=r.get("bound"), frac=r.get("frac"), achieved=r.get("achieved"), unit=r.get("unit"),
                                          avg_us=r.get("avg_us"), avg_us_in_frame=r.get("avg_us_in_frame"), frac_in_frame=r.get("frac_in_frame"),
                                          hbm_frac=(r.get("hbm_view") or {}).get("frac"), traffic=r.get("traffic"),
                                          mfma_busy_frac=r.get("mfma_busy_frac")) if r else None,
                            seconds=round(time.perf_counter() - t0, 1))
        except BaseException as e:  # a reported extra, never the bench line's fate
            out[key] = dict(what=what, value=None, error=f"{type(e).__name__}: {str(e)[:200]}", seconds=round(time.perf_counter() - t0, 1))
    return out


def main():
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)  # `kill -USR1 <pid>` (or timeout -s USR1) dumps the Python stacks of a stuck run
    args = parse()
    _arm_watchdog(args.watchdog)
    ensure_world(args)
    if args.mode == "train":
        return train_main(args)
    if args.mode == "pvrcnn":
        return pvrcnn_main(args)
    if args.mode == "plumbing":
        return plumbing_main(args)
    return forward_main(args)


def forward_main(args):
    from vision3d_amd import dist_util
    rank, local, world = dist_util.env_world()
    assert torch.cuda.is_available(), "bench.py needs a GPU (vision3d_amd has no CPU path)"
    torch.cuda.set_device(local % torch.cuda.device_count())
    pinning = pin_rank_to_gpu_numa(local, world)
    dist_util.init_from_env(BACKEND)  # RCCL; used for the barrier and the max-reduce only (frames are independent)
    import torch.distributed as dist

    from vision3d_amd import synth
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    from vision3d_amd.core.config import second_car_cfg, waymo_range_cfg
    from vision3d_amd.detector import Second

    waymo = args.workload == "waymo"
    cfg = waymo_range_cfg() if waymo else second_car_cfg()
    if args.points is None:
        args.points = 180000 if waymo else 16384
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval().set_precision(args.precision)
    pre = Preprocessor(cfg, seed=0)
    acfg = cfg
    if waymo:  # the reference's anchor grid truncates 150.4/0.4 (fp32) to 375 cells while the BEV map has 376:
        acfg = cfg.clone()  # nudge the bounds so the anchor grid matches the map (throughput stress only)
        acfg.GRID_BOUNDS = [cfg.GRID_BOUNDS[0], cfg.GRID_BOUNDS[1], cfg.GRID_BOUNDS[2], cfg.GRID_BOUNDS[3] + 0.02,
                            cfg.GRID_BOUNDS[4] + 0.02, cfg.GRID_BOUNDS[5]]
    anchors = AnchorGenerator(acfg).anchors.cuda()
    # frame-parallel sharding: rank r owns frames r*B .. r*B+B-1 of the synthetic stream
    make = ((lambda seed, n: synth.make_waymo_cloud(seed, n, order=args.order)) if waymo
            else (lambda seed, n: synth.make_cloud(seed, n, order=args.order)))
    # A stream of N_STREAM different frames per rank (seeds differ per rank and per step), resident in HBM before the timed
    # region; step i runs frames stream[i % N_STREAM]: the timed loop is not a replay of one cache-resident cloud.
    N_STREAM = max(1, args.stream)
    stream_np = [[make((j * world + rank) * args.batch + i, args.points) for i in range(args.batch)] for j in range(N_STREAM)]
    stream = [[torch.from_numpy(c).cuda() for c in frame] for frame in stream_np]
    clouds_np, clouds = stream_np[0], stream[0]
    step_no = [0]

    graphed = None
    if args.path == "graph":
        with torch.no_grad():
            if args.pipeline != 1:
                graphed = model.pipelined_inference(anchors, [c.shape[0] for c in clouds], args.pipeline or MAX_PIPELINE,
                                                    autotune=args.pipeline == 0)
                if args.pipeline == 0:
                    graphed.tune(clouds, args.steps)  # outside warm-up and timed region; timed on windows of --steps frames
            else:
                graphed = model.graphed_inference(anchors, [c.shape[0] for c in clouds])
    last_out = [None]

    def step():
        clouds = stream[step_no[0] % N_STREAM]
        step_no[0] += 1
        with torch.no_grad():
            if graphed is not None:
                r = graphed(clouds)  # pipelined: the oldest finished frame (None while the pipeline fills)
                if r is not None:
                    last_out[0] = r
                return r
            if args.path == "native":
                return model.inference_points(clouds, anchors, dense="mfma")
            item = pre(dict(points=clouds, anchors=anchors))
            return model.inference(item)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipelined = args.pipeline != 1 and args.path == "graph"
    n_ranks_seen = ranks_seen(world, args.gpus)  # the collective really spans --gpus ranks (RCCL)

    def window(src, steps):
        """One timed window: EXACTLY `steps` steps, barrier + synchronize on both sides, pipeline empty at its start; the frames
        still in flight after the last submit are collected inside the window (steps == frames)."""
        nonlocal stream
        keep, stream = stream, src
        if pipelined:
            graphed.flush()
        fence()
        t0 = time.perf_counter()
        o = None
        for _ in range(steps):
            o = step()
        if pipelined:
            rest = graphed.flush()
            o = rest[-1] if rest else last_out[0]
        fence()
        dt = time.perf_counter() - t0
        stream = keep
        return dt, o

    def windows(src, steps, count):
        """`count` back-to-back windows; returns (per-window seconds as max over ranks, last output)."""
        ts, o = [], None
        for _ in range(count):
            dt, o = window(src, steps)
            ts.append(dt)
        if world > 1:
            t = torch.tensor(ts, dtype=torch.float64, device=REDUCE_DEVICE)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts = [float(x) for x in t.tolist()]
        return ts, o

    for _ in range(args.warmup):
        out = step()
    first, out = windows(stream, args.steps, 1)  # also sizes the window count (same on every rank: max-reduced)
    n_win = max(5, min(args.windows, int(20.0 / max(first[0], 1e-6)))) if args.windows > 1 else 1
    rest_t, out2 = windows(stream, args.steps, n_win - 1) if n_win > 1 else ([], None)
    out = out2 if out2 is not None else out
    win_t = np.sort(np.asarray(first + rest_t))
    elapsed = float(np.median(win_t))
    frames = world * args.steps * args.batch
    value = frames / elapsed
    spread = dict(windows=int(len(win_t)), value_p10=frames / float(np.percentile(win_t, 90)),
                  value_p90=frames / float(np.percentile(win_t, 10)), value_min=frames / float(win_t[-1]),
                  value_max=frames / float(win_t[0]), value_first_window=frames / first[0])
    # SURVEY 8(d) timing variant "with H->D copy": the same windows with every step's cloud copied from PINNED host memory into
    # the slot's static buffer on the slot's stream (256 KB per KITTI frame); never part of `value`
    with_h2d = None
    if args.path == "graph" and not args.no_h2d:
        pinned = [[torch.from_numpy(c).pin_memory() for c in frame] for frame in stream_np]
        h_t, _ = windows(pinned, args.steps, max(3, min(n_win, 11)))
        h_med = float(np.median(h_t))
        with_h2d = dict(value=frames / h_med, unit="frames/s", ms_per_step=1e3 * h_med / args.steps, windows=len(h_t),
                        bytes_per_frame=int(sum(c.nbytes for c in stream_np[0])),
                        note="pinned host cloud -> device static buffer (async copy on the frame's stream) inside every step")
    # one frame at a time through the same captured graph (latency view of the same work), not part of `value`: every frame timed on
    # its own (host clock around submit + collect, the frame's 8-byte read is the synchronisation), median / p10 / p90 of >= 200
    single_ms, single_stats = None, None
    if args.path == "graph":
        g1 = graphed.slots[0] if pipelined else graphed
        with torch.no_grad():
            for i in range(10):
                g1(stream[i % N_STREAM])
            torch.cuda.synchronize()
            lat = []
            for i in range(max(200, args.single_frames)):
                s0 = time.perf_counter()
                g1(stream[i % N_STREAM])
                lat.append(time.perf_counter() - s0)
            torch.cuda.synchronize()
        lat = 1e3 * np.sort(np.asarray(lat))
        single_ms = float(np.median(lat))
        single_stats = dict(frames=int(len(lat)), median_ms=single_ms, p10_ms=float(np.percentile(lat, 10)),
                            p90_ms=float(np.percentile(lat, 90)), mean_ms=float(lat.mean()),
                            note="per-frame host clock: cloud copy + one graph launch + the 8-byte result read")

    # ---- per-kernel timing of the sparse backbone with HIP events on the launch stream (rank 0)
    roofline, stages, roofline_dense = None, None, None
    if rank == 0 and not args.no_roofline:
        # One eager pass captures the exact operands of the 14 sparse-conv launches of this frame; every launch is then
        # re-issued REP times back to back on the launch stream inside one HIP-event bracket -- the REP calls are captured
        # in a HIP graph, so there is no interpreter time between them and the average is the kernel's own duration.
        import vision3d_amd.spconv.conv as convmod
        orig = convmod.sparse_conv_forward
        captured = []

        def capture(features, weight, rb, scale=None, shift=None, relu=False, algo=0, packed=None, variant=0, precision="bf16x3"):
            captured.append((features, weight, rb, scale, shift, relu, algo, packed))
            return orig(features, weight, rb, scale, shift, relu, algo, packed, variant, precision)
        convmod.sparse_conv_forward = capture
        with torch.no_grad():  # the per-op sparse backbone (Second.inference(item) itself runs the fused plan: nothing to capture there)
            it = pre(dict(points=clouds, anchors=anchors))
            model.cnn(it["voxel_mean"], it["coordinates"], it["batch_size"])
        convmod.sparse_conv_forward = orig
        REP, layers = 25, []
        side = torch.cuda.Stream()
        from vision3d_amd import _lib as L
        from vision3d_amd.runtime import act_entry_from_tensor, rows_split
        f16s = args.precision == "fp32"
        n_packed = 0
        last_packable = max(i for i, c in enumerate(captured) if c[0].shape[1] >= 16 and c[1].shape[-1] % 16 == 0)
        for li, (features, weight, rb, scale, shift, relu, algo, packed) in enumerate(captured):
            cin_, cout_ = weight.shape[-2], weight.shape[-1]
            packable = features.shape[1] >= 16 and cout_ % 16 == 0
            if packable:
                # The layer's kernel ALONE, in the arithmetic of the run and in the form the pipelined frames run it (a plan in
                # throughput mode): from the second packed layer on the gathered rows are the producer's PRE-SPLIT copy, and a layer
                # that feeds another packed layer writes its rows only in that form.  f16s scale entries are computed once, outside the
                # timed launches (in the frame they are calibrated table entries, not launches).
                packed = convmod.pack_sparse_weight(weight.reshape(-1, cin_, cout_).contiguous(), rb.nbr.shape[0], cin_, cout_, args.precision)
                entry = act_entry_from_tensor(features) if f16s else None
                feat_c = features.contiguous()
                out_buf = torch.empty((rb.n, cout_), dtype=torch.float32, device=features.device)
                L.check(L.lib().v3d_sparse_conv_fwd_packed(L.ptr(feat_c), L.ptr(packed), L.ptr(rb.nbr), L.ptr(rb.n_dev), rb.cap,
                                                            rb.nbr.shape[0], cin_, cout_, L.ptr(scale), L.ptr(shift), int(bool(relu)),
                                                            L.ptr(out_buf), int(rb.n), L.PRECISIONS[args.precision], L.ptr(entry),
                                                            None, None, None, None, L.stream_ptr()), "sparse_conv_fwd_packed2")
                in_s = rows_split(feat_c, args.precision, entry) if n_packed > 0 else None
                feeds_packed = li != last_packable
                out_s = torch.empty((rb.n, 2 * cout_), dtype=torch.int16, device=features.device) if feeds_packed else None
                next_entry = act_entry_from_tensor(out_buf) if (f16s and feeds_packed) else None
                n_packed += 1

                def launch(feat_c=feat_c, packed=packed, rb=rb, scale=scale, shift=shift, relu=relu, out_buf=out_buf, entry=entry,
                           cin_=cin_, cout_=cout_, in_s=in_s, out_s=out_s, next_entry=next_entry):
                    L.check(L.lib().v3d_sparse_conv_fwd_packed(None if in_s is not None else L.ptr(feat_c), L.ptr(packed), L.ptr(rb.nbr),
                                                                L.ptr(rb.n_dev), rb.cap, rb.nbr.shape[0], cin_, cout_, L.ptr(scale),
                                                                L.ptr(shift), int(bool(relu)), None if out_s is not None else L.ptr(out_buf),
                                                                int(rb.n), L.PRECISIONS[args.precision], L.ptr(entry), L.ptr(next_entry),
                                                                None, L.ptr(in_s), L.ptr(out_s), L.stream_ptr()), "sparse_conv_fwd_packed2")
            else:
                def launch(features=features, weight=weight, rb=rb, scale=scale, shift=shift, relu=relu, algo=algo):
                    orig(features, weight, rb, scale, shift, relu, algo, None)
            graph = torch.cuda.CUDAGraph()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                launch()  # warm-up outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad(), torch.cuda.graph(graph):
                for _ in range(REP):
                    launch()
            ts = []
            for trial in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                if trial:
                    ts.append(e0.elapsed_time(e1) * 1e-3 / REP)
            cin, cout = weight.shape[-2], weight.shape[-1]
            st = dict(cin=cin, cout=cout, K=rb.nbr.shape[0], n_in=features.shape[0], n_out=rb.n,
                      pairs=int((rb.nbr[:, :rb.n] >= 0).sum().item()), t_avg_us=1e6 * float(np.mean(ts)))
            st["bytes"] = layer_algorithmic_bytes(st)
            st["gbs"] = st["bytes"] / (st["t_avg_us"] * 1e-6) / 1e9
            layers.append(st)
        # the launches of the dominant kernel: KITTI bs = 1 -> the 7 3x3x3 64->64 layers (all <= 16 384 rows: LDS-ring kernel);
        # Waymo range -> the 3x3x3 64->64 layers with >= 32 768 live rows (64-row LDS-shared-weights kernel)
        big = waymo or args.batch > 1
        dom = [l for l in layers if l["cin"] == 64 and l["cout"] == 64 and l["K"] == 27 and (l["n_out"] >= 32768 if big else l["n_out"] <= 16384)]
        if not dom:
            dom = [l for l in layers if l["cin"] == 64 and l["cout"] == 64 and l["K"] == 27]
        dom_bytes = float(np.mean([l["bytes"] for l in dom]))
        dom_t = float(np.mean([l["t_avg_us"] for l in dom])) * 1e-6
        achieved = dom_bytes / dom_t / 1e9
        # measured copy roofline of THIS box ("vendor peaks measured, not assumed"): 512 MiB device copy
        src = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
        # ... and the same copy AT THE KERNEL'S SIZE: A_min bytes moved (half read, half written) by back-to-back device copies -- the HBM
        # roof a launch of this size can reach at all (a 4 MB launch is over before the memory system is up to speed: the roof as a
        # function of size)
        n_small = max(1024, int(dom_bytes / 2 / 4))
        copy_at_size_us = None
        try:  # 50 copies captured in ONE graph: dependent launches like the layers of a frame, no host time between them
            gcopy = torch.cuda.CUDAGraph()
            side_c = torch.cuda.Stream()
            side_c.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_c):
                for _ in range(3):
                    dst[:n_small].copy_(src[:n_small])
            torch.cuda.current_stream().wait_stream(side_c)
            torch.cuda.synchronize()
            with torch.cuda.graph(gcopy):
                for _ in range(50):
                    dst[:n_small].copy_(src[:n_small])
            gcopy.replay()
            torch.cuda.synchronize()
            c0.record()
            for _ in range(4):
                gcopy.replay()
            c1.record()
            torch.cuda.synchronize()
            copy_at_size_us = c0.elapsed_time(c1) * 1e3 / 200
            del gcopy
        except Exception:  # (a reported extra)
            copy_at_size_us = None
        copy_at_size_gbs = (2 * n_small * 4 / (copy_at_size_us * 1e-6) / 1e9) if copy_at_size_us else None
        del src, dst
        # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed summary of the
        # two rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/r01_h_pmc_traffic.txt) is
        # attached when the workload is the one it was collected on, else null
        DOM_KERNEL = "spconv_fwd_rows_kouter<64,64>" if (big and dom[0]["n_out"] >= 32768) else "spconv_fwd_rows_ring<64,64>"
        traffic, traffic_src = None, None
        pmc_path = os.path.join(REPO, "profiles", "pmc_traffic_waymo.json" if waymo else "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.batch == 1 and args.points == (180000 if waymo else 16384):
            pmc = json.load(open(pmc_path))
            traffic, traffic_src = pmc[DOM_KERNEL]["traffic_bytes"] if DOM_KERNEL in pmc else None, pmc["source"]
        # the same kernel against the matrix pipe: MFMAs it ISSUES (16-row tiles incl. the padded ones of the last workgroup x
        # 27 offsets x Cin/32 x Cout/16 x 3 split terms, 16 384 flop each) and the useful flops (2 Cin Cout per rulebook pair).
        # The offset-outer kernel walks 32-row tiles over 28 offset steps (27 padded to even); the ring kernel takes 2 / 3 / 4
        # sixteen-row tiles per workgroup (csrc/spconv.hip launch_rows: rows + 10 % inside one round of 256 workgroups)
        kouter = DOM_KERNEL.startswith("spconv_fwd_rows_kouter")

        def ring_tiles(n):
            want = n + n // 10
            per = 2 if (kouter or want <= 32 * 256) else (3 if want <= 48 * 256 else 4)
            return per * ((n + 16 * per - 1) // (16 * per)), per
        dom_tiles = float(np.mean([ring_tiles(l["n_out"])[0] for l in dom]))
        tiles_per_wg = sorted({ring_tiles(l["n_out"])[1] for l in dom})
        mfma_issued = dom_tiles * (28 if kouter else 27) * 2 * 4 * 3 * 16384  # Cin/32 = 2, Cout/16 = 4, 3 split terms (csrc/spconv.hip SPC_TERMS)
        mfma_useful = float(np.mean([l["pairs"] for l in dom])) * 2 * 64 * 64
        mfma_alg = 3.0 * mfma_useful  # what fp32-class products cost on the bf16 pipe at best: 3 terms per useful product, no zero rows
        mfma_view = dict(issued_tflops=mfma_issued / dom_t / 1e12, frac_issued=mfma_issued / dom_t / 1e12 / 2500.0,
                         useful_tflops=mfma_useful / dom_t / 1e12, algorithmic_tflops=mfma_alg / dom_t / 1e12,
                         frac_algorithmic=mfma_alg / dom_t / 1e12 / 2500.0, peak_tflops=2500.0,
                         note="dense 16-bit MFMA peak (bf16 = f16 rate); 3 terms per product (lo*Wh + hi*Wl + hi*Wh); 'issued' includes the zero rows "
                              "of absent neighbours, 'algorithmic' = 3 x useful (the floor of a split-precision product on this pipe)")
        # Which roof binds?  The time the launch would take at 100 % of each: A_min at 8 TB/s against the algorithmic matrix work at
        # the dense bf16 peak.  At 64 -> 64 the matrix pipe is the tighter one: `bound` names it and achieved / peak /
        # frac follow it; the HBM view (the figure BASELINE.json's metric quotes) stays beside it.
        t_hbm, t_mfma = dom_bytes / (HBM_PEAK_GBS * 1e9), mfma_alg / 2500e12
        hbm_view = dict(achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, floor_us=t_hbm * 1e6,
                        peak_measured_copy=copy_gbs, frac_of_measured=achieved / copy_gbs,
                        copy_at_size=(dict(bytes=2 * n_small * 4, us_per_copy=copy_at_size_us, gbs=copy_at_size_gbs, frac_of_it=achieved / copy_at_size_gbs,
                                           note="device copies moving the same A_min bytes (half read, half written), 50 dependent launches in one "
                                                "captured graph like the layers of a frame: the rate ANY launch of this size gets out of the memory "
                                                "system (at KITTI size it is over before the memory system is up to speed; at Waymo-range size source "
                                                "and destination sit in the 256 MB last-level cache -- not an HBM figure there)") if copy_at_size_us else None))
        # the same kernel INSIDE the frame (every launch follows a different kernel; row counts of the other frames of the stream):
        # rocprofv3 --kernel-trace --stats of the one-frame-at-a-time run, committed under profiles/ (tools/closing_artifacts.sh)
        in_frame = None
        csv_path = os.path.join(REPO, "profiles", "in_frame_kernel_stats_waymo.csv" if waymo else "in_frame_kernel_stats.csv")
        if os.path.exists(csv_path) and args.batch == 1:
            import csv as _csv
            tag = "spconv_fwd_rows_kouter<64, 64" if kouter else "spconv_fwd_rows_ring<64, 64"
            rows = [r for r in _csv.DictReader(open(csv_path)) if tag in r["Name"]]
            calls = sum(int(r["Calls"]) for r in rows)
            if calls:
                us = sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3
                in_frame = dict(avg_us=us, hbm_frac=dom_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                mfma_frac_algorithmic=mfma_alg / (us * 1e-6) / 2500e12, calls=calls,
                                source="profiles/" + os.path.basename(csv_path))
        mfma_pmc = mfma_busy_from_profiles("waymo" if waymo else "kitti", "spconv_fwd_rows_kouter<64, 64" if kouter else "spconv_fwd_rows_ring<64, 64", dom_t * 1e6) \
            if args.batch == 1 else None
        if t_mfma >= t_hbm:
            roofline = dict(bound="mfma", achieved=mfma_alg / dom_t / 1e12, peak=2500.0, unit="TFLOP/s", frac=mfma_alg / dom_t / 2500e12,
                            floor_us=t_mfma * 1e6)
        else:
            roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS, floor_us=t_hbm * 1e6)
        roofline.update(kernel=DOM_KERNEL, tiles_per_workgroup=tiles_per_wg,
                        tiles_note="isolated timing and in-frame statistics: the latency form (2 / 3 tiles per workgroup by row count); the "
                                   "pipelined graphs that produce `value` run the plan's throughput mode (4 tiles per workgroup: less "
                                   "CU-time per launch, ~20 % longer launches, same bits)" if not kouter else None,
                        launches_per_frame=len(dom), bytes_per_launch=dom_bytes,
                        avg_us=dom_t * 1e6, avg_us_in_frame=in_frame["avg_us"] if in_frame else None,
                        frac_in_frame=(in_frame["mfma_frac_algorithmic" if t_mfma >= t_hbm else "hbm_frac"] if in_frame else None),
                        in_frame=in_frame, traffic=traffic, traffic_source=traffic_src, hbm_view=hbm_view, mfma_view=mfma_view,
                        mfma_busy_frac=(mfma_pmc or {}).get("mfma_busy_frac"), mfma_counters=mfma_pmc,
                        bound_note=f"floor at 100 % of each roof: HBM {t_hbm * 1e6:.2f} us (A_min at 8 TB/s), matrix pipe {t_mfma * 1e6:.2f} us "
                                   "(3 bf16 terms x useful flops at 2.5 PFLOP/s); avg_us = isolated back-to-back launches of each layer, "
                                   "avg_us_in_frame = rocprofv3 average inside the one-frame-at-a-time graph")
        # the other large kernel of the frame: the 3x3 RPN convolution (MFMA-bound).  Algorithmic flops = 2*M*Cout*9*Cin;
        # the kernel issues 3 bf16 MFMA terms per product (split precision), so `issued` = 3x `achieved`.
        from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
        ny, nx = anchors.shape[2:4]
        cdim = cfg.PROPOSAL.C_IN
        xh, xl = to_split_nhwc(torch.randn(args.batch, cdim, ny, nx, device="cuda"), args.precision)
        img = pack_conv_weight(torch.randn(cdim, cdim, 3, 3, device="cuda") / (9 * cdim) ** 0.5, None, args.precision)
        bz = torch.zeros(cdim, device="cuda")
        # (f16s: output entry for magnitudes up to 2^5 -- the unit-variance products of this microbenchmark stay far below)
        pr_d = (xh.v3d_entry, torch.tensor([256.0, 1.0 / 256.0, 128.0, 0.0], device="cuda"), None) if args.precision == "fp32" else None
        for _ in range(3):
            conv2d_split(xh, xl, img, bz, True, cdim, cdim, 3, out_split=True, out_nchw=False, pr=pr_d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv2d_split(xh, xl, img, bz, True, cdim, cdim, 3, out_split=True, out_nchw=False, pr=pr_d)
        e1.record()
        torch.cuda.synchronize()
        t_dense = e0.elapsed_time(e1) * 1e-3 / 20
        fl = 2.0 * args.batch * ny * nx * cdim * cdim * 9
        full_map = dict(kernel="conv2d_bf16x3_large_kernel<3,9>", flops_per_launch=fl, avg_us=t_dense * 1e6, achieved=fl / t_dense / 1e12,
                        issued=3 * fl / t_dense / 1e12, frac=fl / t_dense / 1e12 / 2500.0, frac_issued=3 * fl / t_dense / 1e12 / 2500.0,
                        note="every tile convolved, random dense input (NOT what the frame runs: the upper bound of the matrix work)")
        # ... and the kernel the FRAME runs: the background-skipping 2-D tile form on this frame's sparse map.  Live tiles per layer from
        # the plan's own occupancy bitmap (a 5 x 16-pixel tile of layer i is live when an occupied BEV pixel lies within i + 1 pixels of
        # it: csrc/dense_conv.hip dl_tile2d), matrix work of a live tile = 5 row blocks x 8 column blocks x 9 taps x 4 k-steps x 3 terms
        # = 4 320 MFMAs of 16 384 flop; duration = the in-frame rocprofv3 average (committed csv) and, beside it, the whole dense head of
        # this frame timed with events (7 launches in a captured graph).
        roofline_dense = dict(bound="mfma", kernel="conv2d_bf16x3_tile2d_kernel", full_map=full_map)
        try:
            with torch.no_grad():
                plan_d, flat_d, offs_d = model._plan_for(clouds)
                hi_d, lo_d = plan_d.forward_split(flat_d, offs_d)
                occ_w = plan_d.bev_occupancy(len(clouds)).clone()
                bits = ((occ_w.view(len(clouds), ny, -1, 1) >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1).reshape(len(clouds), ny, -1)[:, :, :nx]
                occ_map = (bits == 0).float().unsqueeze(1)  # inverted bitmap: 0 = occupied
                live, tiles_total = [], len(clouds) * ((ny + 4) // 5) * ((nx + 15) // 16)
                for reach in range(1, 7):
                    near = torch.nn.functional.max_pool2d(occ_map, 2 * reach + 1, 1, reach)
                    pad = torch.nn.functional.pad(near, (0, (-nx) % 16, 0, (-ny) % 5))
                    live.append(int((torch.nn.functional.max_pool2d(pad, (5, 16), (5, 16)) > 0).sum().item()))
                dense, st_d = model.dense_plan(), model.dense_plan().new_state(hi_d.device) if model.skip_background else None
                run_head = lambda: dense.forward(hi_d, lo_d, occ=plan_d.bev_occupancy(len(clouds)) if model.skip_background else None, work=st_d,
                                                 in_entry=plan_d.bev_entry(), range_flag=plan_d.overflow_any())
                for _ in range(3):
                    run_head()
                torch.cuda.synchronize()
                gd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gd):
                    for _ in range(10):
                        run_head()
                tsd = []
                for trial in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); gd.replay(); e1.record(); torch.cuda.synchronize()
                    if trial:
                        tsd.append(e0.elapsed_time(e1) * 1e3 / 10)
            head_us = float(np.mean(tsd))
            in_frame_us, calls_d = None, 0
            csv_d = os.path.join(REPO, "profiles", "in_frame_kernel_stats_waymo.csv" if waymo else "in_frame_kernel_stats.csv")
            if os.path.exists(csv_d) and args.batch == 1:
                import csv as _csv
                rows_d = [r for r in _csv.DictReader(open(csv_d)) if "conv2d_bf16x3_tile2d_kernel" in r["Name"]]
                calls_d = sum(int(r["Calls"]) for r in rows_d)
                if calls_d:
                    in_frame_us = sum(float(r["TotalDurationNs"]) for r in rows_d) / calls_d / 1e3
            mf_tile = 5 * 8 * 9 * 4 * 3
            issued_fl = float(np.mean(live)) * mf_tile * 16384.0
            t_launch = (in_frame_us or head_us / 7.0) * 1e-6
            roofline_dense.update(
                launches_per_frame=6, tiles_total=tiles_total, live_tiles_per_layer=live, live_tiles_mean=float(np.mean(live)),
                compute_units=256, mfma_per_live_tile=mf_tile, flops_issued_per_launch=issued_fl, flops_useful_per_launch=issued_fl / 3.0,
                avg_us=t_launch * 1e6, avg_us_source=("profiles/" + os.path.basename(csv_d) + " (in the frame)") if in_frame_us else "dense head of this frame / 7 launches",
                dense_head_us_this_frame=head_us, issued=issued_fl / t_launch / 1e12, achieved=issued_fl / 3.0 / t_launch / 1e12, peak=2500.0, unit="TFLOP/s",
                frac_issued=issued_fl / t_launch / 1e12 / 2500.0, frac=issued_fl / 3.0 / t_launch / 1e12 / 2500.0,
                mfma_counters=mfma_busy_from_profiles("waymo" if waymo else "kitti", "conv2d_bf16x3_tile2d_kernel", t_launch * 1e6) if args.batch == 1 else None,
                sustained_peak_measured=dict(zero_operands=2200.0, random_operands=1650.0, source="profiles/r06_mfma_chain.txt (tools/mb_mfma_chain.hip: "
                                             "nothing but v_mfma_f32_16x16x32_f16 on all 256 CUs: 7.5 / 9.4-10.3 ns per instruction and SIMD)"),
                note="one frame at a time: one live tile per CU and launch, the launch lasts one tile's chain (occupancy test, 64-request "
                     "neighbourhood load, 1 080 MFMAs per SIMD ~ 9-10 us at the sustained rate, epilogue) and CUs without a live tile idle "
                     "-- live_tiles_mean of 256.  The workgroup is 4 waves x 256 VGPRs + 64 KB of LDS (round 6), so with frames in flight "
                     "the tiles of two frames' launches share a CU (`value`: 3 750 -> 4 090 frames/s on one box)")
        except Exception as e:  # a reported extra
            roofline_dense["error"] = f"{type(e).__name__}: {str(e)[:200]}"
        tot_bytes = sum(l["bytes"] for l in layers)
        tot_t = sum(l["t_avg_us"] for l in layers) * 1e-6
        stages = dict(sparse_conv_launches=len(layers), sparse_conv_us=tot_t * 1e6, sparse_conv_algorithmic_MB=tot_bytes / 1e6,
                      sparse_conv_gbs=tot_bytes / tot_t / 1e9,
                      layers=[{k: (round(v, 2) if isinstance(v, float) else v) for k, v in l.items()} for l in layers])
        # SURVEY 8(d) timing variants: the fused backbone plan of one frame (points -> BEV planes) WITH the voxelizer and the
        # rulebook build, and the same frame's convolutions on the rulebooks that forward left in the plan (prebuilt); each as a
        # captured graph of REP frames between two events
        try:
            with torch.no_grad():
                plan, flat, offsets = model._plan_for(clouds)
                plan.forward_split(flat, offsets)
                torch.cuda.synchronize()

                def graph_us(fn, rep=10):
                    g = torch.cuda.CUDAGraph()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        fn()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g):
                        for _ in range(rep):
                            fn()
                    ts = []
                    for trial in range(4):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        g.replay()
                        e1.record()
                        torch.cuda.synchronize()
                        if trial:
                            ts.append(e0.elapsed_time(e1) * 1e3 / rep)
                    return float(np.mean(ts))
                t_full = graph_us(lambda: plan.forward_split(flat, offsets))
                plan.forward_split(flat, offsets)
                t_reuse = graph_us(lambda: plan.forward_reuse_split(len(offsets) - 1, flat.device))
            stages.update(backbone_us_with_voxelizer_and_rulebook_build=t_full, backbone_us_rulebooks_prebuilt=t_reuse,
                          backbone_note="fused plan, one frame at a time, points -> split BEV planes; 'prebuilt' = only the sparse "
                                        "convolutions and .dense() on the site lists / neighbour tables the previous call left behind")
        except Exception as e:  # a reported extra
            stages.update(backbone_timing_error=repr(e)[:200])

    # ---- the other arithmetic beside the headline: the same pipeline on a second model in bf16x3 (scale-free, 2^-17 per product),
    # a few windows of the same definition; every rank takes part (the windows hold barriers)
    fast_mode = None
    if args.precision == "fp32" and args.path == "graph" and not args.no_fast_mode:
        keep_graphed = graphed
        try:
            torch.manual_seed(0)
            model_f = Second(cfg).cuda().eval().set_precision("bf16x3")
            with torch.no_grad():
                if pipelined:
                    graphed = model_f.pipelined_inference(anchors, [c.shape[0] for c in clouds], args.pipeline or MAX_PIPELINE,
                                                          autotune=args.pipeline == 0)
                    if args.pipeline == 0:
                        graphed.tune(clouds, args.steps)
                else:
                    graphed = model_f.graphed_inference(anchors, [c.shape[0] for c in clouds])
            for _ in range(args.warmup):
                step()
            f_t, _ = windows(stream, args.steps, max(3, min(n_win, 9)))
            f_med = float(np.median(f_t))
            g1f = graphed.slots[0] if pipelined else graphed
            with torch.no_grad():
                for i in range(10):
                    g1f(stream[i % N_STREAM])
                torch.cuda.synchronize()
                latf = []
                for i in range(200):
                    s0 = time.perf_counter()
                    g1f(stream[i % N_STREAM])
                    latf.append(time.perf_counter() - s0)
            fast_mode = dict(value=frames / f_med, unit="frames/s", ms_per_step=1e3 * f_med / args.steps, windows=len(f_t),
                             single_frame_ms=1e3 * float(np.median(latf)), pipeline_tuning=(graphed.tuned if pipelined else None),
                             dtype="bf16x3: bf16 hi + lo pieces, 3 MFMA terms, fp32 accumulate, scale-free; relative product error 2^-17 "
                                   "(strict elementwise error <= 3e-3 on entries above 1e-3 of a layer's maximum)")
            if pipelined:
                graphed.flush()
            graphed = keep_graphed
            del model_f
        except Exception as e:  # a reported extra
            fast_mode = dict(value=None, error=f"{type(e).__name__}: {str(e)[:200]}")
            graphed = keep_graphed

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_baseline = run_cpu_baseline(model, cfg, anchors, make, args, "waymo" if waymo else "kitti")
        except Exception as e:  # the baseline is a reported extra: never lose the bench line over it
            cpu_baseline = dict(value=None, unit="frames/s", error=f"{type(e).__name__}: {str(e)[:200]}")

    if rank == 0:
        wl = ("SECOND forward, bs=1, 180000-pt synthetic Waymo-range sweep per GPU, 0.05 m voxels over +-75.2 m "
              "(BASELINE configs[4])") if waymo else ("SECOND (VoxelNet spconv backbone + BEV head) forward, bs=1, 16384-pt "
                                                      "synthetic KITTI-range cloud per GPU (BASELINE configs[1])")
        line = dict(metric=("frames/sec SECOND fwd, 180k-pt Waymo-range sweep" if waymo else "frames/sec SECOND fwd, 16k-pt KITTI cloud"), value=value, unit="frames/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                    value_is="median over `windows` back-to-back timed windows of `steps` steps each (each window: barrier + "
                             "synchronize on both sides, empty pipeline at its start, max over ranks)",
                    scaling="weak", vs_baseline=None, dtype=DTYPE_NOTE[args.precision], data="synthetic", precision=args.precision,
                    fast_mode=fast_mode,
                    config=dict(workload=wl,
                                frames_per_gpu_per_step=args.batch, points_per_frame=args.points, point_order=args.order,
                                distinct_frames_in_timed_loop=N_STREAM,
                                parallelism=f"frame-parallel replicas x{world}", host_pinning=pinning, pipeline_depth=(graphed.depth if pipelined else 1),
                                pipeline_tuning=(graphed.tuned if pipelined else None),
                                frames_queued_per_stream=(graphed.QUEUE if pipelined else 1),
                                path={"graph": "native backbone plan + split-precision MFMA dense head + device proposal stage, one HIP graph per "
                                               "frame" + (f", {graphed.depth} frames executing on {graphed.depth} streams, the next frame of each stream "
                                                          f"queued behind it ({graphed.capacity} submitted, results collected in order)" if pipelined else ""),
                                      "native": "native backbone plan + split-precision MFMA dense head",
                                      "eager": "eager python -> C ABI"}[args.path]),
                    **spread, with_h2d=with_h2d, n_ranks_seen=n_ranks_seen,
                    single_frame_ms=single_ms, single_frame=single_stats,
                    frames_per_s_one_at_a_time=(world * args.batch * 1e3 / single_ms) if single_ms else None,
                    roofline=roofline, cpu_baseline=cpu_baseline, roofline_dense=roofline_dense, stages=stages,
                    n_proposals=int(out[0].shape[0]))
        if world == 1 and not waymo and args.batch == 1 and not args.no_extra:
            del graphed
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            line["extra"] = run_extras()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
