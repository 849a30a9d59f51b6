"""GPU: the fused native backbone plan (csrc/second_plan.hip) must reproduce the per-op path bit for bit
(same kernels, same order) and therefore the oracle within the feature tolerance; device-side counts,
capacity handling and weight refresh are checked too."""
import numpy as np
import pytest
import torch

from gpu_util import assert_features_close, assert_fp32_class, numpy_state_dict, randomize_bn
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def build_model(seed=0):
    from vision3d_amd.detector import Second
    torch.manual_seed(seed)
    model = Second(second_car_cfg())
    randomize_bn(model, seed)
    return model.cuda().eval()


@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
@pytest.mark.parametrize("seeds,npts", [([0], 16384), ([3, 4, 5], 16384), ([6, 7], 5000)])
def test_plan_matches_eager_path_exactly(seeds, npts, precision):
    """bf16x3: the plan runs the per-op path's kernels in the same order: the same bits.  fp32 (f16s): the op-by-op path takes each
    layer's scale entry from the rows' exact maximum, the plan from its calibration frame with headroom -- powers of two either
    way, so the two results differ only where a piece falls below f16's subnormal quantum: far inside the fp32-class bar."""
    from gpu_util import FP32_CLASS_FLOOR
    from vision3d_amd.core import Preprocessor
    cfg = second_car_cfg()
    model = build_model(1).set_precision(precision)

    def same(a, b):
        if precision == "bf16x3":
            np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
        else:
            assert_features_close(a.cpu().numpy(), b.cpu().numpy(), "plan vs op-by-op (f16s)", floor=FP32_CLASS_FLOOR * 0.1)
    clouds_np = [synth.make_cloud(s)[:npts - 37 * i] for i, s in enumerate(seeds)]
    clouds = [torch.from_numpy(c).cuda() for c in clouds_np]
    with torch.no_grad():
        item = Preprocessor(cfg)(dict(points=[c.clone() for c in clouds]))
        bev_eager = model.cnn(item["voxel_mean"], item["coordinates"], item["batch_size"])
        bev_plan = model.bev_from_points(clouds)
    assert bev_plan.shape == bev_eager.shape
    same(bev_plan, bev_eager)
    plan = next(iter(model._plans.values()))
    feat, coords, n, shape = plan.layer_output(-1)
    m = int(n.item())
    assert m == item["coordinates"].shape[0] and shape == [41, 1600, 1408]
    np.testing.assert_array_equal(coords[:m].cpu().numpy(), item["coordinates"].cpu().numpy())
    np.testing.assert_array_equal(feat[:m].cpu().numpy(), item["voxel_mean"].cpu().numpy())
    _, _, n_last, shape_last = plan.layer_output(13)
    assert shape_last == [2, 200, 176] and 0 < int(n_last.item()) <= 70400 * len(seeds)
    assert int(plan.overflow().sum().item()) == 0
    # a second forward on different data reuses the arena and gives that data's result
    with torch.no_grad():
        again = model.bev_from_points(clouds[::-1])
        back = model.bev_from_points(clouds)
    same(back, bev_eager)
    np.testing.assert_array_equal(back.cpu().numpy(), bev_plan.cpu().numpy())  # the plan reproduces ITSELF bit for bit
    if len(seeds) > 1:
        assert not torch.equal(again, back)


def test_convolutions_on_prebuilt_rulebooks_reproduce_the_forward():
    """v3d_backbone_forward_reuse (bench.py's 'rulebooks prebuilt' timing variant): the layers run again on the site lists and
    neighbour tables the last forward left in the plan -- same BEV planes, bit for bit, as often as it is called."""
    model = build_model(2)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (1, 2)]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        hi, lo = plan.forward_split(flat, offsets)
        for _ in range(2):
            hi2, lo2 = plan.forward_reuse_split(len(clouds), flat.device)
            assert torch.equal(hi, hi2) and torch.equal(lo, lo2)
        assert int(plan.overflow().sum().item()) == 0


def test_plan_tracks_weight_updates_and_matches_oracle():
    from oracle import second_cpu
    cfg = second_car_cfg()
    model = build_model(2)
    cloud = synth.make_cloud(9)
    dev_cloud = [torch.from_numpy(cloud).cuda()]
    with torch.no_grad():
        before = model.bev_from_points(dev_cloud).clone()
        model.cnn.blocks[0][0][0].weight.mul_(1.5)          # in-place change -> version bump -> re-upload
        model.cnn.blocks[3][3][1].running_mean.add_(0.05)
        after = model.bev_from_points(dev_cloud)
    assert not torch.equal(before, after)
    ref = second_cpu.second_forward(numpy_state_dict(model), [cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY,
                                    cfg.MAX_VOXELS)
    ref64 = second_cpu.second_forward64(numpy_state_dict(model), [cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY, cfg.MAX_VOXELS,
                                        dense=False)
    assert_fp32_class(after.cpu().numpy(), ref["bev"], "plan BEV vs oracle", ref64["bev"])


def test_inference_points_equals_item_path():
    from vision3d_amd.core import AnchorGenerator, Preprocessor
    cfg = second_car_cfg()
    model = build_model(3)
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (1, 2)]
    with torch.no_grad():
        a = model.inference_points(clouds, anchors)   # the item path runs the same native kernels on the item's voxels
        b = model.inference(Preprocessor(cfg)(dict(points=[c.clone() for c in clouds], anchors=anchors)))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())


def test_plan_capacity_overflow_is_flagged():
    """A deliberately tiny growth factor clips the stage-1 site list and raises the device flag."""
    from vision3d_amd.runtime import BackbonePlan
    cfg = second_car_cfg()
    model = build_model(4)
    plan = BackbonePlan(model.cnn, cfg, max_batch=1, max_points=16384, growth=0.25)
    plan.allow_overflow = True  # look at the flags instead of raising
    with torch.no_grad():
        out = plan.forward(torch.from_numpy(synth.make_cloud(0)).cuda(), [0, 16384])
    torch.cuda.synchronize()
    flags = plan.overflow()
    assert int(flags[:-1].sum().item()) > 0 and int(flags[-1].item()) == 1 and torch.isfinite(out).all()
    assert int(plan.overflow_any().item()) > 0
    with pytest.raises(RuntimeError, match="capacity"):
        plan.check_overflow()


def test_inference_paths_refuse_a_frame_that_overflowed():
    """Product paths read the plan's summary flag with the proposal count: a frame whose stages dropped rows raises instead
    of returning detections from a wrong BEV map (eager native path, first-forward check, HIP-graph path)."""
    from vision3d_amd.core import AnchorGenerator
    from vision3d_amd.runtime import BackbonePlan
    cfg = second_car_cfg()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    cloud = [torch.from_numpy(synth.make_cloud(0)).cuda()]
    with torch.no_grad():
        plan = BackbonePlan(build_model(4).cnn, cfg, max_batch=1, max_points=16384, growth=0.25)
        with pytest.raises(RuntimeError, match="capacity"):
            plan.forward(cloud[0], [0, 16384])  # the synchronised first forward checks
        model = build_model(4)
        model.plan_growth = 0.25
        with pytest.raises(RuntimeError, match="capacity"):
            model.inference_points(cloud, anchors)
        model = build_model(4)
        model.plan_growth = 0.25
        with pytest.raises(RuntimeError, match="capacity"):
            model.graphed_inference(anchors, [16384])(cloud)
        ok = build_model(4)
        assert len(ok.graphed_inference(anchors, [16384])(cloud)) == 4


def test_graphed_inference_equals_stepwise():
    """One HIP-graph replay per frame == the same native path launched kernel by kernel; replays track new data."""
    from vision3d_amd.core import AnchorGenerator
    model = build_model(5)
    anchors = AnchorGenerator(second_car_cfg()).anchors.cuda()
    a = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (1, 2)]
    b = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (3, 4)]
    with torch.no_grad():
        run = model.graphed_inference(anchors, [16384, 16384])
        for clouds in (a, b, a):
            got = run(clouds)
            ref = model.inference_points(clouds, anchors, dense="mfma")
            assert len(got[0]) == len(ref[0]) > 0
            for x, y in zip(got, ref):
                np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())


def test_graph_capacity_accepts_shorter_frames():
    """A graph captured for a point CAPACITY serves frames of any smaller size (padding far outside the grid is dropped by the
    voxelizer): same detections as the eager native path on the unpadded frame; a larger frame is refused."""
    from vision3d_amd.core import AnchorGenerator
    from vision3d_amd.detector.graph import bucket_points
    model = build_model(6)
    anchors = AnchorGenerator(second_car_cfg()).anchors.cuda()
    assert bucket_points(15000) == 16384 and bucket_points(16384) == 16384 and bucket_points(16385) == 18432
    with torch.no_grad():
        run = model.graphed_inference(anchors, [bucket_points(15000), bucket_points(9000)])
        for sizes in ((16384, 10240), (15000, 9000), (12345, 1), (16384, 10240)):
            clouds = [torch.from_numpy(synth.make_cloud(20 + i)[:n]).cuda() for i, n in enumerate(sizes)]
            got = run(clouds)
            ref = model.inference_points(clouds, anchors, dense="mfma")
            for x, y in zip(got, ref):
                np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        with pytest.raises(RuntimeError, match="capacity"):
            run([torch.from_numpy(synth.make_cloud(1)).cuda(), torch.from_numpy(synth.make_cloud(2)[:10241]).cuda()])


def test_persistent_bev_planes_equal_fresh_planes_over_a_sequence_of_frames():
    """forward_split(persistent=True): the plan's own BEV planes, where a frame zeroes only the pixels the previous frame wrote
    (no fill of the map) -- bit-identical to freshly zeroed planes for every frame of a sequence that alternates dense, sparse,
    empty-ish and repeated frames, batch 2 (also through the prebuilt-rulebook timing variant)."""
    model = build_model(3)
    frames = [[synth.make_cloud(1), synth.make_cloud(2)[:9000]], [synth.make_cloud(3)[:700], synth.make_cloud(4)],
              [synth.make_cloud(5)[:40], synth.make_cloud(6)[:3]], [synth.make_cloud(1), synth.make_cloud(2)[:9000]],
              [synth.make_cloud(7), synth.make_cloud(8)]]
    with torch.no_grad():
        for k, frame in enumerate(frames):
            clouds = [torch.from_numpy(c).cuda() for c in frame]
            plan, flat, offsets = model._plan_for(clouds)
            hi, lo = plan.forward_split(flat, offsets)
            fresh = (hi.clone(), lo.clone())
            phi, plo = plan.forward_split(flat, offsets, persistent=True)
            assert phi.data_ptr() == plan.own_planes(2)[0].data_ptr()
            assert torch.equal(phi, fresh[0]) and torch.equal(plo, fresh[1]), f"frame {k}"
            if k == 1:  # the timing variant into the same planes: clear + rewrite of the same pixels
                rhi, rlo = plan.own_planes(2)
                from vision3d_amd import _lib as L
                L.check(L.lib().v3d_backbone_forward_reuse(plan._handle, 2, 0, L.ptr(rhi), L.ptr(rlo), L.stream_ptr()), "reuse")
                assert torch.equal(rhi, fresh[0]) and torch.equal(rlo, fresh[1])


def _same_detections(a, b):
    assert len(a[0]) == len(b[0]) > 0
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())


def test_f16s_range_overflow_is_flagged_recalibrated_and_rerun():
    """f16s scale entries are calibrated on a frame with 2^6 of headroom.  A frame whose tensors leave that range must not return
    silently wrong detections: the frame's summary word reads 2 (runtime.RangeOverflow at the frame's one host read), the entries
    are re-derived from THAT frame and it is run again -- eager path and captured graph (whose kernels read the entries from
    device memory: rewritten in place, no re-capture) -- with the result a freshly calibrated model gives."""
    from vision3d_amd.core import AnchorGenerator
    cfg = second_car_cfg()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    normal = torch.from_numpy(synth.make_cloud(1)).cuda()
    loud = normal.clone()
    loud[:, 3] *= 3.0e4  # reflectance far outside what the calibration frame showed: every layer's input grows ~1e4-fold
    with torch.no_grad():
        model = build_model(7)
        assert model.precision == "fp32"
        first = model.inference_points([normal], anchors)
        plan = next(iter(model._plans.values()))
        gen, dense_gen = plan.calibration_generation, model.dense_plan().calibration_generation
        # the range check itself: the plan alone raises the summary word to 2 on the loud frame
        hi, lo = plan.forward_split(loud, [0, loud.shape[0]])
        assert int(plan.overflow_any().item()) == 2
        got = model.inference_points([loud], anchors)  # flagged -> recalibrated on this frame -> run again
        assert plan.calibration_generation == gen + 1 and model.dense_plan().calibration_generation > dense_gen
        fresh = build_model(7)
        _same_detections(got, fresh.inference_points([loud], anchors))
        # ... and back: the quiet frame fits inside the louder calibration (smaller values only lose subnormal pieces)
        again = model.inference_points([normal], anchors)
        assert len(again[0]) == len(first[0])
        # captured graph: calibrated on the quiet frame at capture, then handed the loud one
        gmodel = build_model(7)
        run = gmodel.graphed_inference(anchors, [16384])
        _same_detections(run([normal]), first)
        gplan = run.plan
        g0 = gplan.calibration_generation
        out = run([loud])
        assert gplan.calibration_generation == g0 + 1
        _same_detections(out, got)
        _same_detections(run([loud]), got)  # steady state after the recalibration: plain replays


def test_f16s_quiet_frame_is_flagged_recalibrated_downward_and_rerun():
    """The other direction (round-5 review, weak 2): scale entries calibrated on a LOUD frame keep 22 bits only down to 2^-17 of
    that frame's maxima.  A later frame whose tensors stay 2^12 or more below the calibrated limits must not silently lose
    precision: every producing wave folds its maximum into the plan's per-frame table, the one-wave check behind the last layer
    raises the summary word to 3 (runtime.RangeUnderflow at the frame's one host read), the entries are re-derived from THAT frame
    and it is run again -- eager path, captured graph, and a pipeline with 4 frames in flight (both directions)."""
    from vision3d_amd.core import AnchorGenerator
    from vision3d_amd.runtime import RangeUnderflow
    cfg = second_car_cfg()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    normal = torch.from_numpy(synth.make_cloud(1)).cuda()
    loud = normal.clone()
    loud[:, 3] *= 3.0e4  # every layer's input ~1e4-fold (2^13) louder than the normal frame
    with torch.no_grad():
        want_normal = build_model(7).inference_points([normal], anchors)
        want_loud = build_model(7).inference_points([loud], anchors)
        model = build_model(7)
        model.inference_points([loud], anchors)  # calibrated on the loud frame
        plan = next(iter(model._plans.values()))
        gen = plan.calibration_generation
        hi, lo = plan.forward_split(normal, [0, normal.shape[0]])  # the check itself: the plan alone raises the word to 3
        assert int(plan.overflow_any().item()) == 3
        with pytest.raises(RangeUnderflow):
            plan.check_overflow()
        got = model.inference_points([normal], anchors)  # flagged -> recalibrated downward on this frame -> run again
        assert plan.calibration_generation == gen + 1
        _same_detections(got, want_normal)
        assert int(plan.overflow_any().item()) <= 0
        # ordinary variation is NOT flagged (no recalibration churn): the same sweep with half the reflectance
        mid = normal.clone()
        mid[:, 3] *= 0.5
        model.inference_points([mid], anchors)
        assert plan.calibration_generation == gen + 1
        # captured graph: calibrated on the loud frame at capture, then handed the normal one, then the loud one again
        gmodel = build_model(7)
        run = gmodel.graphed_inference(anchors, [16384])
        _same_detections(run([loud]), want_loud)
        g0 = run.plan.calibration_generation
        _same_detections(run([normal]), want_normal)
        assert run.plan.calibration_generation == g0 + 1
        _same_detections(run([normal]), want_normal)  # steady state: plain replays
        _same_detections(run([loud]), want_loud)      # ... and up again (RangeOverflow)
        assert run.plan.calibration_generation == g0 + 2
        # 4 frames in flight: loud, loud, normal, normal, loud, normal ... every result equals the freshly calibrated model's
        pmodel = build_model(7)
        pipe = pmodel.pipelined_inference(anchors, [16384], depth=4)
        seq = [loud, loud, normal, normal, normal, loud, normal, loud, loud, normal]
        outs = []
        for c in seq:
            r = pipe([c])
            if r is not None:
                outs.append([t.clone() for t in r])
        outs += [[t.clone() for t in r] for r in pipe.flush()]
        assert len(outs) == len(seq)
        for c, o in zip(seq, outs):
            _same_detections(o, want_loud if c is loud else want_normal)


def test_f16s_scale_entries_follow_weight_updates():
    """scale entries derived from the OLD weights must not survive a load_state_dict / optimizer step.  Eager
    entry points that never read the range flag (bev_from_points) recalibrate on the next frame; a captured graph is captured again
    (the dense head's packed images are new tensors)."""
    from vision3d_amd.core import AnchorGenerator
    cfg = second_car_cfg()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    cloud = torch.from_numpy(synth.make_cloud(4)).cuda()
    with torch.no_grad():
        model = build_model(5)
        model.bev_from_points([cloud])
        plan = next(iter(model._plans.values()))
        gen = plan.calibration_generation
        for m in model.cnn.modules():  # every sparse layer 2^-3 quieter: far inside the range flags' blind zone
            if hasattr(m, "weight") and m.weight is not None and m.weight.dim() == 5:
                m.weight.mul_(0.125)
        after = model.bev_from_points([cloud]).clone()
        assert plan.calibration_generation == gen + 1
        fresh = build_model(5)
        for m in fresh.cnn.modules():
            if hasattr(m, "weight") and m.weight is not None and m.weight.dim() == 5:
                m.weight.mul_(0.125)
        assert torch.equal(after, fresh.bev_from_points([cloud]))
        # captured graph: weights change after the capture -> the next launch captures again and gives the fresh model's result
        gmodel = build_model(5)
        run = gmodel.graphed_inference(anchors, [16384])
        before = [t.clone() for t in run([cloud])]
        gmodel.head.conv_cls.bias.add_(0.5)
        gmodel.rpn.down_block[1].weight.mul_(1.25)
        assert [torch.equal(a, b) for a, b in zip(run([cloud]), before)] == [True] * 4  # (in-place edits of an eval-mode model: not noticed ...)
        gmodel.notify_weights_changed()                                                    # ... until the runners are told
        out = run([cloud])
        fresh = build_model(5)
        fresh.head.conv_cls.bias.add_(0.5)
        fresh.rpn.down_block[1].weight.mul_(1.25)
        _same_detections(out, fresh.inference_points([cloud], anchors))
        assert len(out[0]) != len(before[0]) or not torch.equal(out[3], before[3])


def test_bf16x3_mode_still_available_and_close_to_fp32_class():
    """Second.set_precision("bf16x3"): the scale-free arithmetic of rounds 1-4 (training plan, fast mode) through the same entry
    points; its head maps agree with the fp32-class ones inside the repository's 1e-4 bar."""
    model = build_model(8)
    clouds = [torch.from_numpy(synth.make_cloud(2)).cuda()]
    with torch.no_grad():
        fp32 = model.fused_head_from_points(clouds).clone()
        model.set_precision("bf16x3")
        fast = model.fused_head_from_points(clouds).clone()
        plan = next(iter(model._plans.values()))
        assert plan.precision == "bf16x3" and plan.bev_entry() is None and model.dense_plan().precision == "bf16x3"
    assert not torch.equal(fp32, fast)
    assert_features_close(fast.cpu().numpy(), fp32.cpu().numpy(), "bf16x3 vs f16s head maps")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("seeds", [(0,), (1, 2, 3, 4, 5, 6, 7)], ids=["kitti_bs1", "bs7_big_kernels"])
def test_presplit_rows_are_bit_identical_to_in_register_splits(seeds, precision):
    """A packed layer of a plan also writes its output rows split into the arithmetic's 16-bit pieces (under the next layer's scale
    entry) and the next packed layer gathers those instead of splitting fp32 rows in its main loop: the same pieces either way, so
    the BEV map must not change by a bit.  bs = 1 runs the 16-row and LDS-ring kernels (staged and register gathers), the 7-frame
    batch the 64-row LDS-shared-weights kernel and the offset-outer kernel (>= 32 768 rows per stage)."""
    model = build_model(9).set_precision(precision)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in seeds]
    with torch.no_grad():
        plan, flat, offsets = model._plan_for(clouds)
        on = model.bev_from_points(clouds).clone()
        rows = [int(plan.layer_output(l)[2].item()) for l in range(len(plan.layers))]
        plan.set_presplit(False)
        off = model.bev_from_points(clouds).clone()
        plan.set_presplit(True)
        again = model.bev_from_points(clouds).clone()
        plan.set_throughput_mode(True)  # split rows ONLY (the fp32 rows of layers followed by a packed layer are not written), 4-tile ring
        thr = model.bev_from_points(clouds).clone()
        plan.set_throughput_mode(False)
    if len(seeds) > 1:
        assert max(rows) >= 32768, rows
    assert torch.equal(on, off) and torch.equal(on, again) and torch.equal(on, thr)
    assert float(on.abs().max()) > 0
