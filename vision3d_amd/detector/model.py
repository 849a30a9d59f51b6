"""PV-RCNN (interface of vision3d/detector/model.py:16-85).

Data flow of the pieces that exist upstream (its `forward` raises: stage 2 was never wired, model.py:84-85; here `forward` /
`inference` wire them the evident way -- SURVEY.md 8(f) rank 3 -- and say so, this is the repository's definition):

    points --FPS--> keypoints --------------------------------------------+
    voxels --sparse CNN--> 4 levels of (xyz, features) + BEV map          |
    [raw points, level 0..3] --set abstraction around the keypoints--> per-keypoint features, + bilinear BEV lookup
    BEV map --ProposalLayer--> (P_cls, P_reg)

Sub-module names (`pnets`, `roi_grid_pool`, `vfe`, `cnn`, `bev`, `proposal_layer`, `refinement_layer`) are the
reference's, so checkpoints load.  FPS / ball query / grouping run on the MI355X kernels behind
vision3d_amd.pointnet2.
"""
import copy

import torch
from torch import nn

from ..pointnet2 import pointnet2_utils as pn2
from ..pointnet2.pointnet2_modules import PointnetSAModuleMSG
from . import layers, proposal, refinement, roi_grid_pool, sparse_cnn


from ..runtime import PlanCache as _PlanCache  # (a dict a deep copy of the model starts empty: events, pinned words, device clones)

_SIDE_STREAMS = {}  # device index -> the stream the keypoint sampling runs on (PV_RCNN.proposal)


def _set_abstraction(radii, mlps, nsamples):
    # the SA module edits its channel lists in place (prepends the xyz channels): hand it a copy of the config's
    return PointnetSAModuleMSG(npoint=-1, radii=radii, nsamples=nsamples, mlps=copy.deepcopy(mlps), use_xyz=True)


class PV_RCNN(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.pnets = self.build_pointnets(cfg)
        self.roi_grid_pool = roi_grid_pool.RoiGridPool(cfg)
        self.vfe = layers.VoxelFeatureExtractor()
        self.cnn = sparse_cnn.CNN_FACTORY[cfg.CNN](cfg)
        self.bev = layers.BEVFeatureGatherer(cfg, self.cnn.voxel_offset, self.cnn.base_voxel_size)
        self.proposal_layer = proposal.ProposalLayer(cfg)
        self.refinement_layer = refinement.RefinementLayer(cfg)

    def build_pointnets(self, cfg):
        """One multi-scale set-abstraction module per feature source (raw points, then the CNN levels)."""
        return nn.Sequential(*(_set_abstraction(radii, mlps, cfg.SAMPLES_PN)
                               for radii, mlps in zip(cfg.PSA.RADII, cfg.PSA.MLPS)))

    def sample_keypoints(self, points):
        """points (B, N, >=3) -> (B, NUM_KEYPOINTS, 3): farthest-point sample of the xyz columns."""
        xyz = points[..., :3].contiguous()
        picked = pn2.furthest_point_sample(xyz, self.cfg.NUM_KEYPOINTS)           # (B, K) int32
        planes = pn2.gather_operation(xyz.transpose(1, 2).contiguous(), picked)   # (B, 3, K)
        return planes.transpose(1, 2).contiguous()

    def _pointnets(self, sources, keypoint_xyz):
        """sources: [(xyz (B, N, 3), features (B, N, C))] -> [(B, C_out, K)], one per source."""
        pooled = []
        for pnet, (xyz, features) in zip(self.pnets, sources):
            _, out = pnet(xyz.contiguous(), None, keypoint_xyz, features_pm=features)  # point-major in: no (B, C, N) round trip
            pooled.append(out)
        return pooled

    def point_feature_extract(self, item, cnn_features, bev_map):
        """Keypoint features: set abstraction over every source concatenated with the BEV lookup (B, C_total, K)."""
        keypoints = item["keypoints"]
        xyz, reflectance = item["points"].split([3, 1], dim=-1)
        sources = [(xyz, reflectance), *cnn_features]
        if self._fused_features_ok(sources, bev_map, keypoints):
            pts = item["points"]
            cloud = pts if (pts.dim() == 3 and pts.shape[2] == 4 and pts.is_contiguous() and pts.dtype == torch.float32) else None
            return self._point_features_fused(sources, bev_map, keypoints, cloud)
        pooled = self._pointnets(sources, keypoints)
        return torch.cat([*pooled, self.bev(bev_map, keypoints)], dim=1)

    # ---- inference: ONE (B, K, C_total) point-major matrix, every set-abstraction scale and the BEV lookup write their column block
    #      (csrc/sa_mlp.hip `ldo`, csrc/pointops.hip v3d_bev_gather_keypoints): no torch.cat per module, per source and at the end, no
    #      transposes to channel-major and back -- RoI-grid pooling gathers point-major rows.  The return value is the (B, C_total, K)
    #      VIEW of that matrix (the reference's layout, model.py:72-74); same values as the op-by-op path.
    def _fused_features_ok(self, sources, bev_map, keypoints):
        return (not torch.is_grad_enabled() and not self.training and keypoints.is_cuda and keypoints.dtype == torch.float32
                and all(p._fusable(f) for p, (_, f) in zip(self.pnets, sources)) and self.bev._native(bev_map, keypoints))

    def _point_features_fused(self, sources, bev_map, keypoints, cloud=None):
        """`cloud`: the (B, N, 4) points whose columns 0-2 / 3 are sources[0]'s xyz / its one feature channel (when contiguous)."""
        b, k = keypoints.shape[:2]
        widths = [sum(p.out_channels()) for p in self.pnets] + [bev_map.shape[1]]
        feats = torch.empty((b, k, sum(widths)), dtype=torch.float32, device=keypoints.device)
        # the ball-query grids of every database of the frame in ONE launch: the five sources, and the keypoints themselves for
        # the RoI-grid pooling that follows (handed over on the returned tensor: RoiGridPool.forward looks for it)
        xyzs = [xyz.contiguous() for xyz, _ in sources]
        kp = keypoints.contiguous()
        grids = None
        if pn2.BALL_QUERY_ALGO == "grid":
            grids = pn2.ball_query_grids([(x, p.max_radius()) for x, p in zip(xyzs, self.pnets)] + [(kp, self.roi_grid_pool.pnet.max_radius())])
        # one launch for the ball queries of all sources (they all ask around the keypoints), one for their first-layer products
        nbrs = prods = None
        plans = [p.pair_plan() for p in self.pnets] if grids else None
        if grids and len(self.pnets) <= 8 and all(pl is not None and len(p.groupers) == 2 for p, pl in zip(self.pnets, plans)):
            nbrs = pn2.ball_query_pairs_many([(grids[i], xyzs[i], p.groupers[0].radius, p.groupers[0].nsample, p.groupers[1].radius,
                                               p.groupers[1].nsample) for i, p in enumerate(self.pnets)], kp)
            prepped = [p.prep_features(f) if i or cloud is None else cloud for i, (p, (_, f)) in enumerate(zip(self.pnets, sources))]
            if cloud is not None:  # the raw source's reflectance read out of the (B, N, 4) cloud itself (column 3): no padded copy
                plans[0] = self.pnets[0].pair_plan(feat_cols=(3,), width=4)
            prods = pn2.linear_rows_many([(f.reshape(-1, f.shape[2]), pl[0]) for f, pl in zip(prepped, plans)])
            sources = [(x, f) for (x, _), f in zip(sources, prepped)]
        col = 0
        for i, (pnet, (_, features), w) in enumerate(zip(self.pnets, sources, widths)):
            pnet.fused_forward(xyzs[i], features, kp, out_pm=feats[:, :, col:col + w], grid=grids[i] if grids else None,
                               neighbours=nbrs[i] if nbrs else None, plan=plans[i] if prods else False,
                               p_all=prods[i].view(features.shape[0], features.shape[1], -1) if prods else None)
            col += w
        self.bev.gather_point_major(bev_map, kp, out_pm=feats[:, :, col:])
        out = feats.transpose(1, 2)
        if grids:
            out._v3d_keypoint_grid = grids[-1]
        return out

    def proposal(self, item):
        """Stage 1.  Adds keypoints, P_cls, P_reg (and the CNN outputs under `_cnn_features` / `_bev_map` for the
        keypoint-feature stage) to `item`."""
        # The keypoints depend on the raw points alone and their farthest-point sampling is a 2 048-step dependent chain on ONE
        # compute unit (2.5 ms for a 16 384-point cloud, csrc/pointops.hip): it runs on a side stream while the other 255 units do the
        # voxel CNN and the proposal head, and joins before the first consumer (the set abstraction of stage 2).  A caller that
        # already knows the next frame can start its sampling earlier still (`prefetch_keypoints`).
        main = torch.cuda.current_stream(item["points"].device)
        if "keypoints" not in item:
            self.prefetch_keypoints(item)
        native = self._native_item(item)
        if native:                    # eval, no autograd, voxels of the device Preprocessor: the sparse CNN as one native plan
            cnn_features, bev_map = self._native_cnn(item)
        else:
            if "voxel_mean" in item:      # device voxelizer output
                voxel_features = item["voxel_mean"]
            else:                         # reference-style (M, K, C) slots + occupancy
                voxel_features = self.vfe(item["features"], item["occupancy"])
            cnn_features, bev_map = self.cnn(voxel_features, item["coordinates"], item["batch_size"])
        if native and self.native_tail:  # the fused [cls | reg] maps are what the native top-k of stage1_proposals reads
            item["_head_maps"] = self.proposal_layer.native_head(bev_map)
            item["P_cls"], item["P_reg"] = self.proposal_layer.maps_from_fused(item["_head_maps"])
        else:
            item.pop("_head_maps", None)
            item["P_cls"], item["P_reg"] = self.proposal_layer(bev_map)
        item["_cnn_features"], item["_bev_map"] = cnn_features, bev_map
        ready = item.pop("_keypoints_ready", None)
        if ready is not None:
            main.wait_event(ready)
            item["keypoints"].record_stream(main)
        return item

    # ---- the sparse CNN of an inference frame through the backbone plan (csrc/second_plan.hip): voxels in, the four sparse levels
    #      and the BEV map out of ONE native call (rulebooks + 14 layers, ~35 launches issued from C++) instead of the module-by-module
    #      path (a rulebook + a row-count read per stage, a scale entry + a launch per layer).  The levels are views of the plan's
    #      stage buffers: rows [0, n) of the stage's last layer, n read in the frame's one host synchronisation together with the
    #      plan's summary word (capacity / f16s range).  Same sites in the same order as the modules (one rulebook builder); features
    #      equal up to the f16s rounding of a calibrated instead of a per-call scale.
    native_cnn = True

    def _native_item(self, item):
        if not self.native_cnn or self.training or torch.is_grad_enabled() or "voxel_mean" not in item:
            return False
        vm, co = item["voxel_mean"], item["coordinates"]
        return vm.is_cuda and co.is_cuda and co.dtype == torch.int32 and vm.dtype == torch.float32 and vm.shape[0] > 0

    def _backbone_plan(self, batch_size, max_points, slot=0):
        from ..runtime import BackbonePlan, PlanCache
        from ..spconv.conv import _SparseConvBase
        plans = self.__dict__.setdefault("_plans", PlanCache())
        dev = next(self.parameters()).device
        key = (str(dev), int(batch_size), int(max_points), int(slot))  # slot: frames in flight keep their levels in arenas of their own
        # the arithmetic of THIS model's sparse modules (set per instance by set_precision, or by hand on the modules), not the class default
        first = self.__dict__.get("_first_sparse_conv")
        if first is None or first[0] != id(self.cnn):
            first = (id(self.cnn), next((m for m in self.cnn.modules() if isinstance(m, _SparseConvBase)), None))
            self.__dict__["_first_sparse_conv"] = first
        precision = first[1].precision if first[1] is not None else _SparseConvBase.precision
        if key not in plans:
            plans[key] = BackbonePlan(self.cnn, self.cfg, max_batch=batch_size, max_points=max_points, device=dev,
                                      growth=self.__dict__.get("plan_growth", 2.0), precision=precision)
            for other in plans.values():  # one calibration per model (see Second.share_calibration)
                if other is not plans[key] and other.f16s and other._calib == "done":
                    plans[key].copy_calibration(other)
                    break
        plans[key].set_precision(precision)
        return plans[key]

    def set_precision(self, precision):
        """Arithmetic of the native sparse paths of this model (the op-by-op modules and the inference plan): "fp32" (f16s, default)
        or "bf16x3" -- the same switch as Second.set_precision."""
        from .. import _lib as L
        if precision not in L.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}")
        from ..spconv.conv import _SparseConvBase
        for m in self.modules():
            if isinstance(m, _SparseConvBase):
                m.precision = precision
        return self

    def _native_cnn(self, item):
        return self._native_cnn_finish(item, self._native_cnn_launch(item))

    def _native_cnn_launch(self, item, slot=0):
        """Enqueue the plan's forward and the copy of its host words (row counts of the levels + summary flag) into pinned memory;
        nothing is waited for.  -> state for `_native_cnn_finish`."""
        from .. import spconv
        vm, co, b = item["voxel_mean"], item["coordinates"], int(item["batch_size"])
        cap_pts = 1 << max(14, (max(vm.shape[0], 1) - 1).bit_length())
        plan = self._backbone_plan(b, max(cap_pts, b * 16384), slot)
        if not plan.__dict__.get("_tuned"):
            # a plan picks its kernels from the sparsity of the FIRST frame it sees, and the variants differ in the last bits: every
            # slot's plan is tuned on the frame the first one saw, so that a frame's result does not depend on the slot it ran in
            seen = self.__dict__.setdefault("_tune_frames", _PlanCache())
            tkey = (b, max(cap_pts, b * 16384))
            if tkey not in seen:
                seen[tkey] = (vm.clone(), co.clone())
            elif slot != 0:
                plan.forward_voxels(seen[tkey][0], seen[tkey][1], b)
        ends = self.__dict__.get("_stage_ends")  # index of the last layer of every stage in the plan's flat layer list (walked once:
        if ends is None or ends[0] != id(self.cnn):  # the module tree of the CNN is ~60 modules)
            ends, k = [id(self.cnn)], 0
            for stage in self.cnn.blocks:
                k += sum(1 for m in stage.modules() if isinstance(m, spconv.conv._SparseConvBase))
                ends.append(k - 1)
            self.__dict__["_stage_ends"] = ends
        ends = ends[1:]
        bev_map = plan.forward_voxels(vm, co, b)
        outs = [plan.layer_output(e) for e in ends[:-1]]
        words = torch.cat([n for _, _, n, _ in outs] + [plan.overflow_any()])
        pinned = self.__dict__.setdefault("_host_words", _PlanCache())
        key = (slot, words.numel())
        if key not in pinned:
            pinned[key] = (torch.empty(words.numel(), dtype=words.dtype).pin_memory(), torch.cuda.Event())
        host, ready = pinned[key]
        host.copy_(words, non_blocking=True)
        ready.record()
        return dict(plan=plan, ends=ends, outs=outs, bev_map=bev_map, host=host, ready=ready, slot=slot)

    def _native_cnn_finish(self, item, st):
        """The one host read of stage 1 (through ITS event: later work may already be queued on the stream), the level views."""
        from .. import spconv
        vm, co, b = item["voxel_mean"], item["coordinates"], int(item["batch_size"])
        plan = st["plan"]
        for attempt in range(2):
            st["ready"].synchronize()
            host = st["host"].tolist()
            if host[-1] in (2, 3) and attempt == 0:  # an f16s tensor left its calibrated range (up or down): recalibrate on this frame, run it again
                torch.cuda.synchronize(vm.device)  # (frames queued behind this one ran on the old entries: their own flags judge them)
                plan.recalibrate()
                st = self._native_cnn_launch(item, st["slot"])
                item["_stage1_rerun"] = True  # (whoever computed the head from the first pass's BEV map does it again)
                continue
            if host[-1] > 0:  # (a clean frame reads -1: the per-frame 0xFF fill)
                plan.check_overflow()  # raises with the layers that hit their capacity
            break
        outs = st["outs"]
        volumes = [spconv.SparseConvTensor(vm, co, self.cnn.grid_shape, b)]
        volumes += [spconv.SparseConvTensor(f[:n], c[:n], shape, b) for (f, c, _, shape), n in zip(outs, host)]
        points = [self.cnn.to_global(stride, vol) for stride, vol in zip(self.cfg.STRIDES, volumes)]
        return points, st["bev_map"]

    PREFETCH_LANES = 4  # side streams the keypoint sampling of successive frames rotates over

    def prefetch_keypoints(self, item):
        """Start the keypoint sampling of `item` (needs item["points"]) on a side stream and return at once: item["keypoints"] is
        the tensor being filled, item["_keypoints_ready"] the event `proposal` waits for before anything reads it.  Calling this
        for frame i + 1 before `inference` of frame i overlaps the sampling with a whole frame of other work.  Successive calls
        rotate over PREFETCH_LANES side streams: farthest-point sampling is a chain of K dependent steps on ONE compute unit
        (2.48 ms for 2 048 of 16 384 points), so the samplings of frames i + 1, i + 2, ... run side by side on different compute
        units when the caller looks that far ahead (measured in round 6, two ahead: no gain while the main stream is bound by its eager
        launches)."""
        points = item["points"]
        lane = self.__dict__.get("_prefetch_lane", 0)
        self.__dict__["_prefetch_lane"] = (lane + 1) % self.PREFETCH_LANES
        main, side = torch.cuda.current_stream(points.device), self._side_stream(points.device, lane)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            points.record_stream(side)  # (the allocator must not hand the cloud's memory out again while the side stream reads it)
            item["keypoints"] = self.sample_keypoints(points)
            item["_keypoints_ready"] = side.record_event()
        return item

    def prefetch_keypoints_many(self, items):
        """The same for SEVERAL frames at once: farthest-point sampling takes a batch -- one workgroup (one compute unit) per cloud in
        ONE launch, 2.48 ms for all of them instead of 2.48 ms each (side streams of their own do not buy that: five streams are
        more than the four hardware queues).  Frames of one size only; else frame by frame."""
        pts = [it["points"] for it in items]
        if len(items) < 2 or any(p.shape[1:] != pts[0].shape[1:] for p in pts):
            return [self.prefetch_keypoints(it) for it in items]
        lane = self.__dict__.get("_prefetch_lane", 0)
        self.__dict__["_prefetch_lane"] = (lane + 1) % self.PREFETCH_LANES
        main, side = torch.cuda.current_stream(pts[0].device), self._side_stream(pts[0].device, lane)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for p in pts:
                p.record_stream(side)
            kp = self.sample_keypoints(torch.cat(pts, dim=0))
            ready = side.record_event()
            k = 0
            for it, p in zip(items, pts):
                it["keypoints"] = kp[k:k + p.shape[0]]
                it["_keypoints_ready"] = ready
                k += p.shape[0]
        return items

    @staticmethod
    def _side_stream(device, lane=0):
        key = (torch.device(device).index, lane)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return _SIDE_STREAMS[key]

    # ---- stage 2 (upstream: `forward` raises, model.py:84-85).  Definition of this repository:
    #   proposals   = per (frame, class) the TOPK highest-scoring anchors of stage 1, decoded (ProposalLayer's own top-k + decode,
    #                 proposal.py:61-77), BEFORE NMS: a fixed (B, n_cls * TOPK, 7) block, which is what RoiGridPool takes;
    #   refinement  = RefinementLayer on the RoI-grid-pooled keypoint features -> 7 residuals + 1 confidence logit per proposal;
    #   refined box = box_encode.decode(residuals, proposal)  (RefinementLayer.apply_refinements);
    #   inference   = refined boxes scored by sigmoid(confidence), rotated NMS per (frame, class) at the stage-1 IoU threshold,
    #                 the per-class score threshold of the config.
    # the proposal stage either side of stage 2 on csrc/proposal.hip (inference frames of the device Preprocessor): top-k + decode in 2
    # launches instead of ~26, refined-box decode + sigmoid + batched rotated NMS + score cut in 3 instead of ~60 -- the main stream of
    # an eager PV-RCNN frame is bound by its launch count.  False: the torch statements below (the cross-check of the tests).
    native_tail = True

    def _native_tail_ok(self, item):
        head = self.proposal_layer
        return (self.native_tail and "_head_maps" in item and not torch.is_grad_enabled()
                and head.native_supported(item["_head_maps"].shape[0], item["anchors"].numel() // (7 * self.cfg.NUM_CLASSES)))

    def stage1_proposals(self, item):
        """-> boxes (B, n_cls * TOPK, 7), scores (B, n_cls * TOPK), class_idx (n_cls * TOPK,) from P_cls / P_reg / anchors."""
        head = self.proposal_layer
        if self._native_tail_ok(item):
            boxes, scores = head.native_topk(item["_head_maps"], item["anchors"])
            class_idx = torch.arange(self.cfg.NUM_CLASSES, device=scores.device).repeat_interleave(head.TOPK)
            return boxes, scores, class_idx
        score_map = item["P_cls"].sigmoid()
        b, n_cls = score_map.shape[:2]
        scores, anchor_idx = score_map.reshape(b, n_cls, -1).topk(head.TOPK, -1)
        boxes = head._decode(item["P_reg"], item["anchors"], anchor_idx)  # (B, n_cls, TOPK, 7)
        class_idx = torch.arange(n_cls, device=scores.device).repeat_interleave(head.TOPK)
        return boxes.reshape(b, -1, head.DOF), scores.reshape(b, -1), class_idx

    def forward(self, item, samples=None, decode=True):
        """Stage 1 + stage 2.  Adds to `item`: keypoints, P_cls, P_reg, keypoint_features (B, 512, K), proposals (B, n, 7),
        proposal_scores (B, n), proposal_class (n,), pooled_features (B, n, 256), R_reg (B, n, 7), R_cls (B, n, 1) and
        boxes_refined (B, n, 7) [decode=False: left to the caller -- `inference` gets them from the native tail].  `samples`
        (B, n, NUM_GRIDPOINTS, 3) in [0, 1) fixes the RoI grid points (the reference draws them with an unseeded torch.rand,
        roi_grid_pool.py:59)."""
        item = self.proposal(item)
        features = self.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        boxes, scores, class_idx = self.stage1_proposals(item)
        pooled = self.roi_grid_pool(boxes, item["keypoints"], features, samples)
        deltas, conf = self.refinement_layer(item["points"], pooled, boxes)
        item.update(keypoint_features=features, proposals=boxes, proposal_scores=scores, proposal_class=class_idx,
                    pooled_features=pooled, R_reg=deltas, R_cls=conf)
        if decode:
            item["boxes_refined"] = self.refinement_layer.apply_refinements(deltas, boxes)
        return item

    # ---- two frames in flight from ONE host thread (round 6).  `inference` enqueues stage 1, WAITS for its row counts (the level views
    # are sized by them), enqueues stage 2 and waits again for the result: the GPU idles while the host enqueues, the host while the
    # GPU drains.  Split in three, stage 1 of frame i + 1 is queued BEFORE frame i's counts are read, and frame i's result is read
    # only after frame i + 1's stage 2 is queued -- the stream never runs dry; every wait is on the frame's own event.  Frames
    # alternate between two plan arenas (`slot`): stage 2 of one reads its levels while stage 1 of the other writes its own.
    #     st = m.inference_begin(item0, 0)
    #     for i ...:  nxt = m.inference_begin(item[i + 1], (i + 1) % 2); h = m.inference_end(st); out = m.inference_collect(prev); prev, st = h, nxt
    def inference_begin(self, item, slot=0):
        """Stage 1 of `item` enqueued (sparse CNN plan + fused head; the keypoint sampling on its side stream if not prefetched);
        nothing is waited for."""
        if not (self._native_item(item) and self.native_tail):
            raise RuntimeError("inference_begin: frames of the device Preprocessor in eval mode without autograd")
        if "keypoints" not in item:
            self.prefetch_keypoints(item)
        st = self._native_cnn_launch(item, slot)
        item["_head_maps"] = self.proposal_layer.native_head(st["bev_map"])
        item["P_cls"], item["P_reg"] = self.proposal_layer.maps_from_fused(item["_head_maps"])
        return dict(item=item, cnn=st)

    def inference_end(self, st, samples=None):
        """Frame `st`: the host read of its stage 1 (its own event), stage 2 and the refinement tail enqueued, the result count on its
        way to pinned memory.  -> handle for `inference_collect`."""
        item = st["item"]
        main = torch.cuda.current_stream(item["points"].device)
        item["_cnn_features"], item["_bev_map"] = self._native_cnn_finish(item, st["cnn"])
        if item.pop("_stage1_rerun", False):
            item["_head_maps"] = self.proposal_layer.native_head(item["_bev_map"])
            item["P_cls"], item["P_reg"] = self.proposal_layer.maps_from_fused(item["_head_maps"])
        ready = item.pop("_keypoints_ready", None)
        if ready is not None:
            main.wait_event(ready)
            item["keypoints"].record_stream(main)
        features = self.point_feature_extract(item, item["_cnn_features"], item["_bev_map"])
        boxes, scores, class_idx = self.stage1_proposals(item)
        pooled = self.roi_grid_pool(boxes, item["keypoints"], features, samples)
        deltas, conf = self.refinement_layer(item["points"], pooled, boxes)
        item.update(keypoint_features=features, proposals=boxes, proposal_scores=scores, proposal_class=class_idx,
                    pooled_features=pooled, R_reg=deltas, R_cls=conf)
        item["boxes_refined"], raw = self.proposal_layer.native_refine_nms(deltas, boxes, conf, finalize=False)
        slot = st["cnn"]["slot"]
        pinned = self.__dict__.setdefault("_host_count", _PlanCache())
        if slot not in pinned:
            pinned[slot] = (torch.empty(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
        host, done = pinned[slot]
        host.copy_(raw[4], non_blocking=True)
        done.record()
        return dict(raw=raw, host=host, done=done)

    def inference_collect(self, h):
        """-> (boxes, batch_idx, class_idx, scores) of the frame behind handle `h` (waits for ITS event only)."""
        h["done"].synchronize()
        n = int(h["host"][0])
        return [t[:n] for t in h["raw"][:4]]

    def inference(self, item, samples=None):
        """-> (boxes (K, 7), batch_idx (K,), class_idx (K,), scores (K,)) by decreasing score, the return contract of
        Second.inference / ProposalLayer.inference."""
        from ..ops import batched_nms_rotated
        if "voxel_mean" in item and self.native_tail and not self.training and not torch.is_grad_enabled():
            item = self.forward(item, samples, decode=False)
            if self._native_tail_ok(item):
                item["boxes_refined"], out = self.proposal_layer.native_refine_nms(item["R_reg"], item["proposals"], item["R_cls"])
                return out
            item["boxes_refined"] = self.refinement_layer.apply_refinements(item["R_reg"], item["proposals"])
        else:
            item = self.forward(item, samples)
        boxes = item["boxes_refined"]
        b, n = boxes.shape[:2]
        scores = item["R_cls"].sigmoid().reshape(-1)
        boxes = boxes.reshape(-1, boxes.shape[-1])
        batch_idx = torch.arange(b, device=boxes.device).repeat_interleave(n)
        class_idx = item["proposal_class"].repeat(b)
        n_cls = self.cfg.NUM_CLASSES
        keep = batched_nms_rotated(boxes[:, [0, 1, 3, 4, 6]].contiguous(), scores, class_idx + n_cls * batch_idx, 0.01)
        boxes, batch_idx, class_idx, scores = (x[keep] for x in (boxes, batch_idx, class_idx, scores))
        mask = self.proposal_layer._above_score_thresh(scores, class_idx)
        return [x[mask] for x in (boxes, batch_idx, class_idx, scores)]
