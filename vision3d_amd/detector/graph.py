"""HIP-graph capture of the SECOND inference path (one replay instead of ~150 launches per frame)."""
import torch


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
        self.graph = None  # captured on the first call, after a warm-up on THAT frame (see _capture)

    def _capture(self):
        """Warm-up on a side stream with the real first frame in the static buffer -- it uploads weights, fills the
        allocator pools and lets the backbone plan pick its kernels from the observed sparsity (BackbonePlan.tune;
        an all-zero buffer would tune for a one-voxel frame) -- then capture."""
        dev = self.static_points.device
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
        hi, lo = self.plan.forward_split(self.static_points, self.offsets)
        head = self.model.head
        maps = self.dense.forward(hi, lo)
        self.native = head.native_supported(len(self.frame_sizes), self.anchors.numel() // (7 * self.model.cfg.NUM_CLASSES))
        if self.native:
            return head.native_proposals(maps, self.anchors)
        return head.proposals_padded(*head.maps_from_fused(maps), self.anchors)

    def load(self, clouds):
        assert len(clouds) == len(self.frame_sizes)
        for c, a, b in zip(clouds, self.offsets[:-1], self.offsets[1:]):
            assert c.shape[0] == b - a, "frame size differs from the captured geometry: re-capture"
            self.static_points[a:b].copy_(c, non_blocking=True)

    def replay(self):
        if self.graph is None:
            self._capture()
        self.graph.replay()
        return self.outputs

    def __call__(self, clouds):
        self.load(clouds)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        head = self.model.head
        return head.finalize_native(*self.outputs) if self.native else head.finalize(*self.outputs)


class PipelinedSecond(object):
    """Throughput mode for independent frames: `depth` GraphedSecond instances (own plan arena, own static buffers, own
    HIP graph) on `depth` streams.  The sparse half of a frame is a chain of small launches that fills a fraction of
    the chip; with two frames in flight it overlaps the other frame's dense head.  Latency per frame is unchanged;
    results come back in submission order.

        run.submit(clouds)        # copy + graph launch on the next slot's stream, returns immediately
        run.collect()             # oldest frame in flight: waits for ITS stream only, -> (boxes, batch, class, scores)
    """

    def __init__(self, model, anchors, frame_sizes, depth=2):
        dev = next(model.parameters()).device
        self.slots = [GraphedSecond(model, anchors, frame_sizes, slot=i) for i in range(depth)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.pending = []  # slot indices in submission order
        self.next_slot = 0

    def submit(self, clouds):
        i = self.next_slot
        assert i not in self.pending, "collect() the oldest frame before reusing its slot"
        g, st = self.slots[i], self.streams[i]
        st.wait_stream(torch.cuda.current_stream())  # the caller's cloud tensors are ready
        with torch.cuda.stream(st), torch.no_grad():
            g.load(clouds)
            if g.graph is None:
                g._capture()
            g.graph.replay()
        self.pending.append(i)
        self.next_slot = (i + 1) % len(self.slots)

    def collect(self):
        i = self.pending.pop(0)
        g = self.slots[i]
        with torch.cuda.stream(self.streams[i]):
            out = g.model.head.finalize_native(*g.outputs) if g.native else g.model.head.finalize(*g.outputs)
        return [t.clone() for t in out]  # the slot's static buffers are overwritten by its next frame

    def __call__(self, clouds):
        """submit this frame, return the oldest finished one once the pipeline is full (None while it fills)."""
        if len(self.pending) == len(self.slots):
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
