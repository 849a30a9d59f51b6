"""Frame-parallel multi-GPU plumbing: one process per GPU, frames are independent units.

The reference is single-process/single-GPU (training.md:6); SURVEY.md section 8(e): inference shards
frames across ranks with NO data-path collective; the only communication is a barrier and a max-reduce
of the elapsed time (bench.py), and -- for the train configuration -- one flat gradient all-reduce.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* when world > 1; returns (rank, local, world)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return rank, local, world


def shard_frames(n_frames_total, rank, world):
    """Contiguous block of frame ids owned by `rank` (frames r*B .. r*B+B-1 for equal shards)."""
    per = (n_frames_total + world - 1) // world
    lo = min(rank * per, n_frames_total)
    return list(range(lo, min(lo + per, n_frames_total)))


def barrier(world):
    if world > 1:
        dist.barrier()


def max_over_ranks(value, world, device="cpu"):
    """MAX-reduce of a python float (the bench contract: step time = slowest rank)."""
    if world == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients_flat(params, world):
    """ONE collective per step: gradients flattened into a single bucket, summed over ranks, divided by the
    world size and scattered back (~1.83 M parameters = 7.3 MB fp32 -- latency-, not
    bandwidth-bound over xGMI, so a single flat bucket beats per-tensor reductions)."""
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return flat.numel()


class TwoPhaseGradReducer(object):
    """The train step's gradient all-reduce in two buckets, the first one overlapped with the backward.

    Autograd reaches the sparse backbone LAST (it is the first thing the forward runs), and the backbone's backward is one
    native call of ~5 ms (runtime.PlanTrainFunction).  When that call starts, every gradient of the dense half (RPN + heads:
    the `early` parameters) has already been accumulated, so their bucket is reduced ASYNCHRONOUSLY while the sparse backward
    runs (`start_early()`, hooked in through `BackbonePlan.pre_backward_hook`); `finish()` reduces the `late` bucket (the
    backbone's own gradients), waits for both and writes the averages back.  Two collectives of ~4 MB and ~3 MB per step instead
    of one of 7 MB after the backward; results are identical to `allreduce_gradients_flat` (sums of the same values).
    """

    def __init__(self, early_params, late_params, world):
        self.early, self.late, self.world = list(early_params), list(late_params), int(world)
        self._work, self._flat, self._grads, self._deferred, self._early_ids = None, None, None, False, set()

    @staticmethod
    def _bucket(params):
        grads = [p.grad for p in params if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads]) if grads else None
        return grads, flat

    @staticmethod
    def _scatter(grads, flat, world):
        flat.div_(world)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    def start_early(self):
        """Call once per step when the early parameters' gradients are complete (idempotent until `finish`).

        Only started when EVERY early parameter that requires a gradient has one: a gradient that arrives later (or is still
        being accumulated) would be left unreduced -- in that case the whole reduction is deferred to `finish()`.
        Stream order: the bucket is built by kernels on the CALLER's stream (torch.cat inside autograd's backward); the
        collective runs on the backend's own stream (RCCL) -- torch's ProcessGroupNCCL makes that stream wait for the caller's
        current stream at enqueue time, which is the ordering this relies on (gloo reduces host-visible tensors synchronously
        with respect to the caller's stream)."""
        if self.world == 1 or self._work is not None:
            return
        if any(p.requires_grad and p.grad is None for p in self.early):
            self._deferred = True
            return
        self._grads, self._flat = self._bucket(self.early)
        self._early_ids = {id(g) for g in self._grads}
        if self._flat is not None:
            self._work = dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """After backward(): reduce the late bucket, complete the early one.  Returns the number of reduced elements."""
        if self.world == 1:
            return 0
        if self._work is None:  # the hook never fired or deferred: plain two-bucket reduction now (all gradients are final)
            self._grads, self._flat = self._bucket(self.early)
            self._early_ids = {id(g) for g in self._grads}
            if self._flat is not None:
                self._work = dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            # an early parameter whose gradient tensor appeared after start_early (it had none then, so it is not in the bucket)
            stragglers = [p for p in self.early if p.grad is not None and id(p.grad) not in self._early_ids]
            if stragglers:
                g2, f2 = self._bucket(stragglers)
                dist.all_reduce(f2, op=dist.ReduceOp.SUM)
                self._scatter(g2, f2, self.world)
        n = 0
        grads, flat = self._bucket(self.late)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            self._scatter(grads, flat, self.world)
            n += flat.numel()
        if self._work is not None:
            self._work.wait()
            self._scatter(self._grads, self._flat, self.world)
            n += self._flat.numel()
        self._work, self._flat, self._grads, self._deferred = None, None, None, False
        return n
