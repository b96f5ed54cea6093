"""Which chain of a captured two-stream section keeps the queue when the hipGraph is replayed (LABNOTES 9.7).

L "layers"; per layer the MAIN chain runs three dependent kernels and a join kernel, the SIDE chain two kernels that start from the
previous join's output and feed the next join -- the shape of a decoder layer (main: z_proj, conv Block, posterior Block,
reparameterise; side: z_feat_proj, prior Block).  Captured twice: with the side chain's first kernel enqueued right after the fork
(the natural way to write it), and with the main chain's next kernel enqueued first and the side stream waiting on an event recorded
at the fork.  Same kernels, same dependencies; prints the replayed time per layer.   usage: python tools/graph_queue_probe.py [L]"""
import sys
import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
n = 1 << 20  # ~4 MB per tensor: kernels of a few microseconds


def build(main_first):
    a, b, c = (torch.zeros(n, device=dev) for _ in range(3))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
            for _ in range(L):
                # fork: the side chain reads `a` (the previous join's output)
                if main_first:
                    ev = torch.cuda.Event(); ev.record(main)
                    a1 = a * 1.0001                      # main chain, kernel 1 -- enqueued BEFORE the side chain's first kernel
                    side.wait_event(ev)
                else:
                    side.wait_stream(main)
                with torch.cuda.stream(side):
                    b1 = a + 1.0
                    b2 = b1 * 0.5
                if not main_first:
                    a1 = a * 1.0001
                a2 = a1 + 0.25
                a3 = a2 * 0.999
                main.wait_stream(side)                    # join
                a = a3 + b2
            c.copy_(a)
    return g, c


for main_first in (False, True):
    g, c = build(main_first)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps / L
    print("%-44s %6.1f us per layer (4 main + 2 side kernels), checksum %.4f" % (
        "main chain first after the fork:" if main_first else "side chain first after the fork:", us, float(c[0])))
