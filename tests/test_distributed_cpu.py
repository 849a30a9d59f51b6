"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py / the train step -- frame sharding, barrier +
max-over-ranks timing, and the single flat gradient all-reduce (sum / world, equal to the mean of the
per-rank gradients)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vision3d_amd import dist_util as D
    r, _, w = D.init_from_env("gloo")
    frames = D.shard_frames(5, r, w)
    D.barrier(w)
    slowest = D.max_over_ranks(1.0 + r, w)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    x = torch.full((4, 6), float(r + 1))
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    n = D.allreduce_gradients_flat(list(model.parameters()), w)
    out[rank] = dict(frames=frames, slowest=slowest, n=n, local=local, reduced=[p.grad.clone() for p in model.parameters()])
    torch.distributed.destroy_process_group()


def test_two_rank_sharding_timing_and_gradient_allreduce():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["frames"] == [0, 1, 2] and b["frames"] == [3, 4]          # contiguous, disjoint, complete
    assert a["slowest"] == b["slowest"] == 2.0                         # MAX over ranks
    assert a["n"] == b["n"] == sum(p.numel() for p in a["local"])      # one flat bucket
    for ga, gb, ra, rb in zip(a["local"], b["local"], a["reduced"], b["reduced"]):
        torch.testing.assert_close(ra, (ga + gb) / 2)
        torch.testing.assert_close(ra, rb)


def _worker_two_phase(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vision3d_amd import dist_util as D
    r, _, w = D.init_from_env("gloo")
    torch.manual_seed(0)
    first = torch.nn.Linear(6, 5)    # "late": its gradients come out of the backward last (the sparse backbone's role)
    second = torch.nn.Linear(5, 3)   # "early": complete when the backward reaches `first`
    reducer = D.TwoPhaseGradReducer(list(second.parameters()), list(first.parameters()), w)
    fired = []

    class Mark(torch.autograd.Function):  # stands where PlanTrainFunction.backward calls plan.pre_backward_hook
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            fired.append(all(p.grad is not None for p in second.parameters()))
            reducer.start_early()
            return g

    x = torch.full((4, 6), float(r + 1))
    second(Mark.apply(torch.relu(first(x)))).sum().backward()
    local = [p.grad.clone() for p in list(first.parameters()) + list(second.parameters())]
    n = reducer.finish()
    reduced = [p.grad.clone() for p in list(first.parameters()) + list(second.parameters())]
    # the hook-less path (module-by-module training) reduces the same two buckets inside finish()
    for p, g in zip(list(first.parameters()) + list(second.parameters()), local):
        p.grad.copy_(g)
    n2 = reducer.finish()
    again = [p.grad.clone() for p in list(first.parameters()) + list(second.parameters())]
    out[rank] = dict(n=n, n2=n2, local=local, reduced=reduced, again=again, fired=fired)
    torch.distributed.destroy_process_group()


def test_two_phase_gradient_reducer_overlapped_bucket_equals_flat_mean():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_two_phase, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["fired"] == [True] and b["fired"] == [True]  # the early bucket's gradients were complete when the hook fired
    assert a["n"] == a["n2"] == sum(p.numel() for p in a["local"])
    for ga, gb, ra, rb, aa in zip(a["local"], b["local"], a["reduced"], b["reduced"], a["again"]):
        torch.testing.assert_close(ra, (ga + gb) / 2)
        torch.testing.assert_close(ra, rb)
        torch.testing.assert_close(aa, ra)


def _worker_two_phase_incomplete(rank, world, port, out):
    """start_early() while an early parameter has NO gradient yet: the early reduction is deferred to finish(), and
    a gradient tensor that appears after an early start is reduced as a straggler -- ranks never diverge."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vision3d_amd import dist_util as D
    r, _, w = D.init_from_env("gloo")
    torch.manual_seed(0)
    e1, e2, late = torch.nn.Linear(4, 4), torch.nn.Linear(4, 2), torch.nn.Linear(3, 4)
    reducer = D.TwoPhaseGradReducer(list(e1.parameters()) + list(e2.parameters()), list(late.parameters()), w)
    x = torch.full((2, 3), float(r + 1))
    # case 1: hook fires when only e2 has gradients (e1's arrive later) -> deferred
    h = e1(late(x))
    h.retain_grad()
    e2(h).sum().backward(inputs=list(e2.parameters()) + [h])
    reducer.start_early()
    deferred = reducer._deferred and reducer._work is None
    h.grad = None
    e2.zero_grad()
    e2(e1(late(x))).sum().backward()
    local = [p.grad.clone() for m in (e1, e2, late) for p in m.parameters()]
    reducer.finish()
    reduced = [p.grad.clone() for m in (e1, e2, late) for p in m.parameters()]
    # case 2: an early parameter gets its FIRST gradient tensor after an (accepted) early start -> straggler pass
    e1.weight.requires_grad_(False)
    for m in (e1, e2, late):
        m.zero_grad(set_to_none=True)
    e2(e1(late(x))).sum().backward()
    reducer.start_early()                      # every early parameter that requires a gradient has one: starts
    started = reducer._work is not None
    e1.weight.requires_grad_(True)
    e1.weight.grad = torch.full_like(e1.weight, float(r + 1))  # appears later
    reducer.finish()
    out[rank] = dict(deferred=deferred, started=started, local=local, reduced=reduced, straggler=e1.weight.grad.clone())
    torch.distributed.destroy_process_group()


def test_two_phase_reducer_defers_incomplete_buckets_and_reduces_stragglers():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_two_phase_incomplete, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["deferred"] and b["deferred"] and a["started"] and b["started"]
    for ga, gb, ra, rb in zip(a["local"], b["local"], a["reduced"], b["reduced"]):
        torch.testing.assert_close(ra, (ga + gb) / 2)
        torch.testing.assert_close(ra, rb)
    torch.testing.assert_close(a["straggler"], torch.full_like(a["straggler"], 1.5))
    torch.testing.assert_close(b["straggler"], a["straggler"])


def test_single_rank_paths_are_noops():
    from vision3d_amd import dist_util as D
    assert D.shard_frames(3, 0, 1) == [0, 1, 2]
    assert D.max_over_ranks(0.25, 1) == 0.25
    assert D.allreduce_gradients_flat([], 1) == 0
    red = D.TwoPhaseGradReducer([], [], 1)
    red.start_early()
    assert red.finish() == 0
