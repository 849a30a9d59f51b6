"""CPU: how `bench.py --gpus N` turns into N ranks (launch_plan) -- the reference has no multi-GPU mode (training.md:6), so the
1/2/4/8-GPU axis of the metric is this repository's.  The 2-rank run itself needs a GPU: tests/test_gpu_bench_launch.py."""
import importlib.util
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("v3d_bench", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_single_rank_needs_no_launcher(bench):
    assert bench.launch_plan(1, {}, ["--steps", "5"], 1, backend="nccl") is None


def test_a_rank_of_the_right_job_runs_in_place(bench):
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, [], 8, backend="nccl") is None


def test_gpus_without_launcher_reexecutes_under_torch_distributed_run(bench):
    cmd = bench.launch_plan(2, {}, ["--gpus", "2", "--steps", "5"], 2, backend="nccl", port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5"]
    assert bench.launch_plan(4, {}, [], 4, backend="nccl")[cmd.index("--master-port") + 1].isdigit()  # a free port is picked


def test_mismatches_fail_loudly(bench):
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        bench.launch_plan(2, {"WORLD_SIZE": "4"}, [], 8, backend="nccl")
    with pytest.raises(SystemExit, match="shows 1 GPU"):
        bench.launch_plan(8, {}, [], 1, backend="nccl")
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, [], 1, backend="nccl")
    assert bench.launch_plan(2, {}, [], 1, backend="gloo") is not None  # gloo: ranks may share a device (control-flow check)


def test_numa_pinning_reads_the_topology_and_degrades_quietly(bench, tmp_path, monkeypatch):
    """One rank per GPU is pinned to the cores of its GPU's NUMA node (bench.pin_rank_to_gpu_numa): the cpulist parser, the sysfs
    walk on a fake tree, and the no-topology case (this container: no GPU, no such files) which must do nothing."""
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert bench.parse_cpulist("") == []

    class Props:
        pci_domain_id, pci_bus_id, pci_device_id = 0, 0x65, 0
    monkeypatch.setattr(bench.torch.cuda, "get_device_properties", lambda i: Props())
    dev = tmp_path / "bus/pci/devices/0000:65:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices/system/node/node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("16-31,144-159\n")
    cpus = bench.numa_cpus_of_gpu(0, sysfs=str(tmp_path))
    assert cpus == list(range(16, 32)) + list(range(144, 160))
    (dev / "numa_node").write_text("-1\n")  # no affinity reported
    assert bench.numa_cpus_of_gpu(0, sysfs=str(tmp_path)) is None
    assert bench.numa_cpus_of_gpu(0, sysfs=str(tmp_path / "missing")) is None
    monkeypatch.delenv("V3D_BENCH_PIN", raising=False)
    assert bench.pin_rank_to_gpu_numa(0, 1) is None  # single-rank jobs are left alone


def test_extra_sub_lines_are_valid_short_runs_of_the_other_configurations(bench):
    """The default run's `extra` object (round 6: the other BASELINE configurations on the driver's clock): every entry's argv parses,
    names a mode bench.py has, and stays a SHORT run; the keys are the ones the round's review asked for."""
    keys = [k for k, _, _ in bench.EXTRA_RUNS]
    assert {"waymo", "plumbing", "pvrcnn_stage2"} <= set(keys) and len(keys) == len(set(keys))
    for key, what, argv in bench.EXTRA_RUNS:
        a = bench.parse(argv + ["--no-cpu-baseline", "--no-fast-mode", "--no-h2d", "--no-extra", "--watchdog", "240"])
        assert a.mode in ("forward", "train", "pvrcnn", "plumbing") and a.no_extra and a.no_cpu_baseline
        assert a.steps <= 100 and a.gpus == 1, key
        assert "configs[" in what or "batch of 8" in what


def test_counter_summaries_attach_to_the_roofline_objects(bench):
    """bench.mfma_busy_from_profiles reads the committed counter passes (profiles/pmc_mfma.json) and divides the busy cycles by the
    CALLER's launch duration; unknown kernels / runs give None."""
    m = bench.mfma_busy_from_profiles("kitti", "conv2d_bf16x3_tile2d_kernel", 20.4)
    assert m is not None and 0.1 < m["mfma_busy_frac"] < 0.5 and m["mfma_instructions_per_launch"] > 1e5
    assert abs(m["mfma_busy_cycles_per_launch"] / m["mfma_instructions_per_launch"] - 16.0) < 0.01  # the 4-pass 16x16x32 instruction
    assert bench.mfma_busy_from_profiles("kitti", "no_such_kernel", 10.0) is None
    assert bench.mfma_busy_from_profiles("kitti", "conv2d_bf16x3_tile2d_kernel", None) is None
    k = bench.mfma_busy_from_profiles("waymo", "spconv_fwd_rows_kouter<64, 64", 47.0)
    assert k is not None and 20.0 < k["sustained_floor_us"] < 30.0  # 2 600 MFMAs per SIMD at 10 ns
