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


def test_single_rank_paths_are_noops():
    from vision3d_amd import dist_util as D
    assert D.shard_frames(3, 0, 1) == [0, 1, 2]
    assert D.max_over_ranks(0.25, 1) == 0.25
    assert D.allreduce_gradients_flat([], 1) == 0
