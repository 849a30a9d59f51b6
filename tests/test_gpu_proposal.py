"""GPU: the fused proposal stage (csrc/proposal.hip) vs (a) the torch statement of proposal.py:61-80 running on the
same device and (b) the CPU oracle (oracle/second_cpu.proposals).  Ties in the score (where torch.topk's order is
unspecified) are avoided by construction in the exact-equality cases and exercised separately through the
documented (score desc, anchor index asc) rule."""
import numpy as np
import pytest
import torch

from gpu_util import dev
from vision3d_amd.core import AnchorGenerator
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


def make_maps(cfg, B, H, W, seed, spread=1.0):
    """Logits ~ N(-3, 1): the top-100 of 70 400 land around sigmoid(1) where fp32 scores of distinct logits stay
    distinct (saturated scores would tie and leave the order to torch.topk's unspecified tie-breaking)."""
    rng = np.random.default_rng(seed)
    n_anchor = cfg.NUM_CLASSES * cfg.NUM_YAW
    cls = (rng.standard_normal((B, n_anchor, H, W)) * spread - 3.0).astype(np.float32)
    reg = (rng.standard_normal((B, n_anchor * 7, H, W)) * 0.3).astype(np.float32)
    return np.concatenate([cls, reg], 1)


def multi_class_cfg():
    cfg = second_car_cfg().clone()
    cfg.ANCHORS = [dict(cfg.ANCHORS[0]), dict(cfg.ANCHORS[0], wlh=[0.6, 0.8, 1.73], center_z=-0.6, score_thresh=0.2),
                   dict(cfg.ANCHORS[0], wlh=[0.6, 1.76, 1.73], center_z=-0.6, score_thresh=0.4)]
    cfg.NUM_CLASSES = 3
    return cfg


@pytest.mark.parametrize("B,multi", [(1, False), (2, False), (2, True)])
def test_native_matches_torch_statement_and_oracle(oracle, B, multi):
    from oracle import second_cpu
    from vision3d_amd.detector.proposal import ProposalLayer
    cfg = multi_class_cfg() if multi else second_car_cfg()
    head = ProposalLayer(cfg).cuda()
    anchors = AnchorGenerator(cfg).anchors
    H, W = anchors.shape[2:4]
    maps = make_maps(cfg, B, H, W, seed=B + 10 * multi)
    n_anchor = cfg.NUM_CLASSES * cfg.NUM_YAW
    sig = torch.from_numpy(maps[:, :n_anchor]).sigmoid().reshape(B, cfg.NUM_CLASSES, -1)
    top = sig.topk(cfg.PROPOSAL.TOPK + 1, -1)[0]
    assert bool((top[..., :-1] > top[..., 1:]).all()), "test data must be tie-free in the top-k"
    out = head.inference_native(dev(maps), anchors.cuda())
    cls_map, reg_map = head.maps_from_fused(dev(maps))
    ref = head.inference_from_maps(cls_map.clone(), reg_map, anchors.cuda())
    assert out[0].shape[0] > 0
    for a, b, name in zip(out, ref, ("boxes", "batch_idx", "class_idx", "scores")):
        assert a.shape == b.shape, name
        if a.dtype == torch.int64:
            assert torch.equal(a, b), name
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=name)
    cpu = second_cpu.proposals(maps[:, :n_anchor], maps[:, n_anchor:], anchors.numpy(), cfg.NUM_CLASSES, cfg.NUM_YAW, 7,
                               cfg.PROPOSAL.TOPK, [a["score_thresh"] for a in cfg.ANCHORS])
    assert out[0].shape[0] == cpu[0].shape[0]
    np.testing.assert_array_equal(out[1].cpu().numpy(), cpu[1])
    np.testing.assert_array_equal(out[2].cpu().numpy(), cpu[2])
    np.testing.assert_allclose(out[0].cpu().numpy(), cpu[0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out[3].cpu().numpy(), cpu[3], rtol=1e-5, atol=1e-7)


def test_many_groups_fall_back_to_the_torch_statement():
    """B * n_cls * TOPK > 1024 candidates exceed the one-workgroup sort: inference_native routes to the op-by-op path."""
    from vision3d_amd.detector.proposal import ProposalLayer
    cfg = multi_class_cfg()
    head = ProposalLayer(cfg).cuda()
    anchors = AnchorGenerator(cfg).anchors
    H, W = anchors.shape[2:4]
    assert head.native_supported(3) and not head.native_supported(4)
    maps = dev(make_maps(cfg, 4, H, W, seed=77))
    out = head.inference_native(maps, anchors.cuda())
    ref = head.inference_from_maps(*head.maps_from_fused(maps), anchors.cuda())
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


def test_candidate_order_with_ties():
    """All-equal and saturated logits: candidates are the lowest anchor indices, in index order; nothing reads
    out of bounds; duplicates of one box collapse to a single keep."""
    from vision3d_amd.detector.proposal import ProposalLayer
    cfg = second_car_cfg()
    head = ProposalLayer(cfg).cuda()
    anchors = AnchorGenerator(cfg).anchors
    H, W = anchors.shape[2:4]
    maps = np.zeros((1, cfg.NUM_YAW * 8, H, W), np.float32)
    maps[:, :cfg.NUM_YAW] = 30.0  # sigmoid saturates to exactly 1.0 everywhere: 70400-way tie
    boxes, bi, ci, scores = head.inference_native(dev(maps), anchors.cuda())
    # zero deltas: box i == anchor i; first TOPK anchors of yaw 0, row 0 are adjacent cells 0.4 m apart -> NMS at
    # IoU 0.01 keeps every box that does not touch an earlier kept one
    assert scores.numel() > 0 and bool((scores == 1.0).all())
    flat = anchors.reshape(-1, 7)[: cfg.PROPOSAL.TOPK].cuda()
    kept = boxes[:, None, :] == flat[None, :, :]
    idx = kept.all(-1).float().argmax(1)
    assert bool(kept.all(-1).any(1).all()), "kept boxes are among the lowest-index anchors"
    assert bool((idx[1:] > idx[:-1]).all()), "ties resolve by ascending anchor index"
    # a single dominant anchor plus a 70 399-way tie below it
    maps[:, :cfg.NUM_YAW] = 2.0
    maps[0, 1, 5, 7] = 5.0
    boxes2, _, _, scores2 = head.inference_native(dev(maps), anchors.cuda())
    np.testing.assert_array_equal(boxes2[0].cpu().numpy(), anchors[0, 1, 5, 7].numpy())


def test_graph_and_eager_native_paths_agree():
    from vision3d_amd import synth
    from vision3d_amd.detector import Second
    cfg = second_car_cfg()
    torch.manual_seed(0)
    model = Second(cfg).cuda().eval()
    anchors = AnchorGenerator(cfg).anchors.cuda()
    clouds = [torch.from_numpy(synth.make_cloud(3)).cuda()]
    with torch.no_grad():
        eager = model.inference_points(clouds, anchors)
        g = model.graphed_inference(anchors, [c.shape[0] for c in clouds])
        rows = []
        for _ in range(4):  # back-to-back replays with only blit copies in between (regression: graph memset nodes
            out = g(clouds)  # lost their 0xFF pattern on the 2nd replay -> full hash tables -> zero voxels)
            rows.append([int(g.plan.layer_output(l)[2].item()) for l in (0, 13)])
    assert rows[0][0] > 10000 and rows[0][1] > 1000 and all(r == rows[0] for r in rows), rows
    for a, b in zip(eager, out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("autotune", [False, True], ids=["two_streams", "autotuned_streams"])
def test_pipelined_frames_match_single_frame_graph(autotune):
    """Throughput mode (frames in flight on separate streams / graphs / plan arenas; fixed depth 2, or depth and streams
    picked by measurement on the first frame): every frame's result equals the one-frame-at-a-time graph result, in
    submission order, for a stream of DIFFERENT clouds of the same size."""
    from test_gpu_dense_conv import build_model  # randomised BatchNorm + larger head weights: outputs depend on the cloud
    from vision3d_amd import synth
    cfg = second_car_cfg()
    model = build_model(3)
    anchors = AnchorGenerator(cfg).anchors.cuda()
    frames = [[torch.from_numpy(synth.make_cloud(s)).cuda()] for s in (1, 2, 3, 4, 5, 6, 7)]
    with torch.no_grad():
        single = model.graphed_inference(anchors, [frames[0][0].shape[0]])
        want = [[t.clone() for t in single(f)] for f in frames]
        pipe = model.pipelined_inference(anchors, [frames[0][0].shape[0]], depth=3 if autotune else 2, autotune=autotune)
        got = []
        for f in frames:
            if len(pipe.pending) < pipe.depth:
                pipe.submit(f)
                continue
            r = pipe.collect()  # the oldest frame: views of its slot's static buffers (collect synchronised that slot's stream)
            copy = [t.clone() for t in r]
            pipe.submit(f)  # goes to ANOTHER slot: the ring has one slot more than frames in flight
            torch.cuda.synchronize()  # the frame just submitted has run to its end ...
            for t, c in zip(r, copy):  # ... contract: a result stays valid until the NEXT collect(), i.e. across this submit
                assert torch.equal(t, c), "collected tensors changed before the next collect()"
            got.append(copy)
        got += [[t.clone() for t in r] for r in pipe.flush()]
    assert len(got) == len(want)
    if autotune:
        assert pipe.tuned is not None and 2 <= pipe.depth <= 3 and pipe.tuned["depth"] == pipe.depth
    distinct = {tuple(w[0].cpu().numpy().round(4).ravel()[:70]) for w in want}
    assert len(distinct) > 1, "test frames must give different proposals"
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a, b)


def test_pipeline_filled_to_capacity_keeps_order_and_results():
    """Round 6: two frames are queued per stream (`capacity` = 2 x depth submitted and not yet collected; the next frame of a stream
    is enqueued before the current one is collected, each frame waited for through its own event, its result words written by the
    last kernel into pinned host memory).  Through `run(clouds)` -- what bench.py calls -- and `collect(copy=True)`: every result
    equals the one-frame-at-a-time graph's, in submission order, for a stream of different clouds longer than the slot ring."""
    from test_gpu_dense_conv import build_model
    from vision3d_amd import synth
    cfg = second_car_cfg()
    model = build_model(3)
    anchors = AnchorGenerator(cfg).anchors.cuda()
    frames = [[torch.from_numpy(synth.make_cloud(s)).cuda()] for s in range(1, 14)]  # 13 frames > capacity + 1 slots
    with torch.no_grad():
        single = model.graphed_inference(anchors, [frames[0][0].shape[0]])
        want = [[t.clone() for t in single(f)] for f in frames]
        pipe = model.pipelined_inference(anchors, [frames[0][0].shape[0]], depth=2)
        assert pipe.capacity == 2 * pipe.depth == 4 and len(pipe.slots) == pipe.capacity + 1
        got, in_flight_max = [], 0
        for f in frames:
            r = pipe(f)  # collects the oldest frame once `capacity` are submitted, then submits
            in_flight_max = max(in_flight_max, len(pipe.pending))
            if r is not None:
                got.append([t.clone() for t in r])
        assert in_flight_max == pipe.capacity
        while pipe.pending:
            got.append(pipe.collect(copy=True))
        torch.cuda.synchronize()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for a, b in zip(g, w):
            assert torch.equal(a, b)
