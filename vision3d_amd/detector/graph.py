"""HIP-graph capture of the SECOND inference path (one replay instead of ~150 launches per frame).

A captured graph has its frame geometry baked in (point capacities are kernel arguments).  Real sweeps vary in size, so a
graph is captured for a CAPACITY per frame and a shorter frame is padded with points far outside every grid: the voxelizer
drops them before they can touch a voxel, a count or an order, so the result is the unpadded frame's bit for bit
(`bucket_points` rounds a size up to the capture quantum; a frame above the capacity needs a graph captured for a larger bucket).
"""
import torch

PAD_COORDINATE = 1.0e30  # outside any GRID_BOUNDS: dropped by the voxelizer's range test (csrc/voxelize.hip)


def bucket_points(n, quantum=2048):
    """Point capacity to capture for frames of about n points: the next multiple of `quantum`."""
    return max(quantum, -(-int(n) // quantum) * quantum)


class GraphedSecond(object):

    def __init__(self, model, anchors, frame_sizes, slot=0):
        self.model, self.anchors = model, anchors
        self.frame_sizes = [int(n) for n in frame_sizes]
        self.offsets = [0]
        for n in self.frame_sizes:
            self.offsets.append(self.offsets[-1] + n)
        dev = next(model.parameters()).device
        c_in = model.cfg.C_IN
        self.static_points = torch.zeros((self.offsets[-1], c_in), dtype=torch.float32, device=dev)
        cap_pts = 1 << max(14, (self.offsets[-1] - 1).bit_length())
        self.plan = model.backbone_plan(len(self.frame_sizes), cap_pts, slot)
        self.dense = model.dense_plan()
        # this slot's tile counters (self-resetting), persistent RPN planes and tile states
        self.work = self.dense.new_state(dev) if model.skip_background else None
        # the frame's two result words (proposal count, summary flag): pinned host memory the LAST kernel of the graph writes itself
        self.host_out = torch.zeros(2, dtype=torch.int32).pin_memory()
        self.done = torch.cuda.Event()  # recorded behind every launch: the frame is waited for through it, not through its stream
                                        # (PipelinedSecond queues the next frame of a stream behind it)
        self.graph = None  # captured on the first call, after a warm-up on THAT frame (see _capture)

    def _capture(self):
        """Warm-up on a side stream with the real first frame in the static buffer -- it uploads weights, fills the
        allocator pools and lets the backbone plan pick its kernels from the observed sparsity (BackbonePlan.tune;
        an all-zero buffer would tune for a one-voxel frame) -- then capture."""
        dev = self.static_points.device
        self.model.share_calibration(self.plan)  # (f16s: a slot captured after a calibrated peer takes its scale entries)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: API calls of other host threads (e.g. the RCCL watchdog of a multi-GPU job polling its events)
        # must not invalidate this capture
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = self._body()

    def _body(self):
        # (the slot's plan keeps its own BEV planes: a frame clears the pixels the previous one wrote, no 18 MB fill)
        hi, lo = self.plan.forward_split(self.static_points, self.offsets, persistent=True)
        head = self.model.head
        maps = self.dense.forward(hi, lo, occ=self.plan.bev_occupancy(len(self.offsets) - 1) if self.model.skip_background else None,
                                  work=self.work, in_entry=self.plan.bev_entry(), range_flag=self.plan.overflow_any())
        self.native = head.native_supported(len(self.frame_sizes), self.anchors.numel() // (7 * self.model.cfg.NUM_CLASSES))
        if self.native:
            return head.native_proposals(maps, self.anchors, self.plan.overflow_any(), host_out=self.host_out)
        return head.proposals_padded(*head.maps_from_fused(maps), self.anchors)

    def load(self, clouds):
        if len(clouds) != len(self.frame_sizes):
            raise RuntimeError(f"graph captured for {len(self.frame_sizes)} frame(s), got {len(clouds)}")
        for c, a, b in zip(clouds, self.offsets[:-1], self.offsets[1:]):
            n = c.shape[0]
            if n > b - a:
                raise RuntimeError(f"frame of {n} points exceeds the captured capacity {b - a}: capture a graph for a larger "
                                   "bucket (detector/graph.py:bucket_points)")
            self.static_points[a:a + n].copy_(c, non_blocking=True)
            if n < b - a:
                self.static_points[a + n:b].fill_(PAD_COORDINATE)

    def weights_changed(self):
        """One integer comparison per launch (Second.notify_weights_changed); the tensors are looked at only when it moved."""
        epoch = self.model.__dict__.get("_weights_epoch", 0)
        if epoch == self.__dict__.get("_seen_epoch"):
            return False
        self._seen_epoch = epoch
        return self.plan.weights_changed() or self.dense.weights_changed()

    def launch(self):
        """Enqueue the frame sitting in the static buffer.  Captures on first use -- and AGAIN when a parameter of the model changed
        since the capture (load_state_dict, an optimizer step): the dense head's packed images are new tensors whose addresses the
        old graph does not know, and the f16s scale entries belong to the old weights' activations; the warm-up passes of the new
        capture upload the weights and recalibrate on this frame."""
        if self.graph is not None and self.weights_changed():
            torch.cuda.synchronize(self.static_points.device)  # nothing may still be running the old graph
            self.graph = None
        if self.graph is None:
            self._capture()
            self._seen_epoch = self.model.__dict__.get("_weights_epoch", 0)
        self.graph.replay()
        self.done.record()

    def replay(self):
        self.launch()
        return self.outputs

    def _finalize(self):
        head = self.model.head
        if self.native:
            return head.finalize_native(*self.outputs, overflow_flag=self.plan.overflow_any(), done=self.done)
        out = head.finalize(*self.outputs)
        if self.plan.f16s:
            self.plan.check_overflow()
        return out

    def finalize(self, peers=()):
        """The frame's host read.  f16s: a frame that left the calibrated range (runtime.RangeOverflow) is recalibrated ON and run
        again -- the scale entries, the empty-map responses and the persistent planes' tile states are device memory the captured
        graph reads, rewritten in place by one eager pass over the frame still sitting in the static buffer.  `peers`: the other
        slots of a pipeline (they share the dense head's entries): the caller has drained them; their tile states are reset too."""
        from ..runtime import RangeOverflow
        try:
            return self._finalize()
        except RangeOverflow:
            dev = self.static_points.device
            torch.cuda.synchronize(dev)  # nothing else may be reading the entries while they are rewritten
            self.plan.recalibrate()
            self.dense.recalibrate()
            with torch.no_grad():
                self._body()  # eager: calibration passes of the plan, then of the dense head, on this frame
            self.model.spread_calibration(self.plan)  # every plan of the model follows (device copies into the tables the graphs read)
            for g in (self,) + tuple(peers):
                g.after_recalibration()
            self.graph.replay()
            self.done.record()
            return self._finalize()

    def after_recalibration(self):
        """The dense head's scale entries changed under this slot's captured graph: nothing in its persistent planes is in place."""
        if self.work is not None and self.work.key is not None:
            self.work.ensure(self.dense, *self.work.key[:3])

    def __call__(self, clouds):
        self.load(clouds)
        self.launch()
        return self.finalize()


def choose_streams(time_of, n_candidates, max_depth, min_gain=0.015):
    """The selection rule of PipelinedSecond.tune, free of any device: `time_of(ids)` -> seconds per frame with the
    candidate streams `ids` (one slot each).  Every pair is timed and the best kept; a further stream is added
    greedily -- the one that gives the shortest time -- as long as that beats the current set by `min_gain`.
    Returns (chosen ids, {depth: best time seen at that depth})."""
    if max_depth < 2 or n_candidates < 2:
        return [0], {}
    pairs = {(a, b): time_of([a, b]) for a in range(n_candidates) for b in range(a + 1, n_candidates)}
    best = min(pairs, key=pairs.get)
    chosen, t_best = list(best), pairs[best]
    log = {2: t_best}
    while len(chosen) < min(max_depth, n_candidates):
        trial = {c: time_of(chosen + [c]) for c in range(n_candidates) if c not in chosen}
        c = min(trial, key=trial.get)
        log[len(chosen) + 1] = trial[c]
        if trial[c] > (1.0 - min_gain) * t_best:  # a deeper pipeline has to pay for its arena
            break
        chosen, t_best = chosen + [c], trial[c]
    return chosen, log


class PipelinedSecond(object):
    """Throughput mode for independent frames: `depth` GraphedSecond instances (own plan arena, own static buffers, own
    HIP graph) on `depth` streams.  The sparse half of a frame is a chain of small launches that fills a fraction of
    the chip; with several frames in flight it overlaps the other frames' dense heads.  Latency per frame is unchanged;
    results come back in submission order.

        run.submit(clouds)        # copy + graph launch on the next slot's stream, returns immediately
        run.collect()             # oldest frame in flight: waits for ITS stream only, -> (boxes, batch, class, scores)

    There is one slot MORE than frames in flight: the tensors `collect()` returns are views of the collected slot's static
    output buffers, and that slot is the last one the ring hands out again -- they stay valid until the NEXT collect()
    (no per-frame clone kernels; `collect(copy=True)` clones for callers that keep results longer).

    Which streams: two HIP streams only overlap if their hardware queues sit on different command-processor pipes,
    and the runtime gives no handle on that (measured on one MI355X with 2 / 3 / 4 streams taken in creation order
    and GPU_MAX_HW_QUEUES = 4, 8, 16: 434 / 530 / 445, 468 / 347 / 523, 483 / 644 / 369 us per frame -- luck of the
    mapping, not depth).  With `autotune=True` (depth = the maximum) the first submitted frame is used to MEASURE: the
    slots are captured, every pair out of 8 candidate streams runs a few frames, the best pair is extended greedily
    while a deeper pipeline still pays, and that stream set is kept.
    """

    CANDIDATE_STREAMS = 8
    TUNE_FRAMES = 8
    QUEUE = 2  # frames queued per stream (see `capacity`; 3 measured slower: 4 390 vs 4 770 frames/s, 1: 4 180)

    def __init__(self, model, anchors, frame_sizes, depth=2, autotune=False):
        dev = next(model.parameters()).device
        self.max_depth = depth
        self.slots = [GraphedSecond(model, anchors, frame_sizes, slot=i) for i in range(depth * self.QUEUE + 1)]
        for g in self.slots:  # frames of different slots share the GPU: kernels chosen for CU-time, not for the shortest launch
            g.plan.set_throughput_mode(True)  # (round 6, two frames queued per stream: 4 757 vs 4 145 frames/s without)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.autotune, self.tuned = bool(autotune), None
        self.pending = []  # (slot, stream index) in submission order
        self.next_slot = self.next_stream = 0

    @property
    def depth(self):
        return len(self.streams)

    @property
    def capacity(self):
        """Frames submitted and not yet collected: QUEUE per stream.  Only `depth` of them EXECUTE at a time (more streams than that
        lose: the hardware runs four queues side by side, a fifth stream is time-sliced -- tune() measures it); the second frame of a
        stream is already enqueued when the first finishes, so the host's share of a frame -- the wake-up from the wait, the
        finalize, the next load and graph launch, ~60-90 us -- no longer stands between two frames of a stream.  Each frame is waited
        for through its own event, not through its stream."""
        return self.depth * self.QUEUE

    # ---- one frame on one slot -------------------------------------------------------------------------------
    def _launch(self, i, stream, clouds):
        g = self.slots[i]
        stream.wait_stream(torch.cuda.current_stream())  # the caller's cloud tensors are ready
        with torch.cuda.stream(stream), torch.no_grad():
            g.load(clouds)
            g.launch()

    def _finish(self, i, stream):
        g = self.slots[i]
        with torch.cuda.stream(stream):
            return g.finalize(peers=[p for p in self.slots if p is not g])

    # ---- stream selection by measurement ---------------------------------------------------------------------
    def _time_streams(self, streams, clouds, frames):
        """seconds per frame of ONE window of `frames` frames on these streams: pipeline empty at its start, the frames still in
        flight collected inside it -- the definition of a timed window of the benchmark (best of three)."""
        import time
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0, inflight = time.perf_counter(), []
            for f in range(frames):
                i = f % len(streams)
                if len(inflight) == len(streams):
                    j = inflight.pop(0)
                    self._finish(j, streams[j])
                self._launch(i, streams[i], clouds)
                inflight.append(i)
            for j in inflight:
                self._finish(j, streams[j])
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / frames
            best = t if best is None else min(best, t)
        return best

    def tune(self, clouds, window_frames=None):
        """Pick the stream set (and depth <= len(slots)) that gives the shortest time per frame on THIS frame.
        window_frames: how many frames the caller's timed windows hold (bench.py: --steps).  The best depth depends on it -- a
        window starts with an empty pipeline and ends when its last frame is collected, so short windows reward a deeper pipeline
        (more frames start at once) while long ones are decided by the steady state, where a fourth frame in flight only adds
        contention (round 4, same box: 20-frame windows 3 705 vs 3 594 frames/s at depth 4 vs 3, 300-frame windows 3 616 vs 3 747).
        The candidates are therefore timed on windows of that length (8 ... 64 frames)."""
        frames = self.TUNE_FRAMES if window_frames is None else max(self.TUNE_FRAMES, min(int(window_frames), 64))
        dev = self.slots[0].static_points.device
        cands = [torch.cuda.Stream(device=dev) for _ in range(self.CANDIDATE_STREAMS)]
        for i in range(len(self.slots)):  # capture every slot (tunes the plans on the real frame), one at a time
            self._launch(i, cands[0], clouds)
            self._finish(i, cands[0])
        chosen, log = choose_streams(lambda ids: self._time_streams([cands[x] for x in ids], clouds, frames), len(cands), self.max_depth)
        self.streams = [cands[x] for x in chosen]
        self.tuned = dict(depth=len(chosen), window_frames=frames, us_per_frame={k: round(v * 1e6, 1) for k, v in log.items()})
        self.pending, self.next_slot, self.next_stream = [], 0, 0
        return self.tuned

    # ---- the pipeline ----------------------------------------------------------------------------------------
    def submit(self, clouds):
        if self.autotune and self.tuned is None:
            self.tune(clouds)
        assert len(self.pending) < self.capacity, "collect() the oldest frame before submitting another one"
        i, j = self.next_slot, self.next_stream
        self._launch(i, self.streams[j], clouds)
        self.pending.append((i, j))
        self.next_slot = (i + 1) % (self.capacity + 1)  # capacity + 1 slots in the ring (tuning may have shortened the stream list)
        self.next_stream = (j + 1) % self.depth

    def collect(self, copy=False):
        """Oldest frame in flight -> [boxes, batch_idx, class_idx, scores].  The tensors alias that slot's static buffers and
        stay valid until the NEXT collect() (the ring has one slot more than frames in flight); copy=True clones them."""
        i, j = self.pending.pop(0)
        out = self._finish(i, self.streams[j])  # synchronises the slot's stream: the data is complete for any consumer
        if not copy:
            return out
        consumer = torch.cuda.current_stream()
        with torch.cuda.stream(self.streams[j]):
            res = [t.clone() for t in out]
        consumer.wait_stream(self.streams[j])
        for t in res:
            t.record_stream(consumer)
        return res

    def __call__(self, clouds):
        """submit this frame, return the oldest finished one once the pipeline is full (None while it fills)."""
        if self.autotune and self.tuned is None:
            self.tune(clouds)
        if len(self.pending) == self.capacity:
            out = self.collect()
            self.submit(clouds)
            return out
        self.submit(clouds)
        return None

    def flush(self):
        outs = []
        while self.pending:
            outs.append(self.collect())
        return outs
