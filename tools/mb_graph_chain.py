"""How fast does the chip retire DEPENDENT kernels of captured graphs?  N streams, one graph each: a chain of K tiny kernels (one
64-thread workgroup each).  Prints us per kernel of one chain alone and the aggregate with 2 / 3 / 4 / 6 chains in flight --
the floor under a launch-bound frame (SECOND bs = 1: 38 dependent launches) however little work a launch holds."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    graphs = []
    for s in streams:
        x = torch.zeros(64, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            for _ in range(3):
                x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(K):
                x.add_(1.0)
        graphs.append((g, x))
    torch.cuda.synchronize()
    for n in (1, 2, 3, 4, 6, 8):
        best = None
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R = 50
            for r in range(R):
                for i in range(n):
                    with torch.cuda.stream(streams[i]):
                        graphs[i][0].replay()
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / (R * n * K)
            best = t if best is None else min(best, t)
        print(f"{n} chains in flight: {1e6 * best:6.2f} us per kernel aggregate ({1e6 * best * n:6.2f} us per kernel per chain)")


if __name__ == "__main__":
    main()
