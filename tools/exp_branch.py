"""Experiment: does running two independent half-batch UNet forwards on two streams (one captured graph) beat one full-batch
forward?  Sizes the gain of batch-window branches for the latency-bound small-map section."""
import copy
import sys
import time
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from UNet import UNetModel  # noqa: E402
from bench import fill_weights, mri_like  # noqa: E402

dev = torch.device("cuda:0")
S = int(os.environ.get("S", 256))
m = UNetModel(S, 128, n_heads=2, attention_resolutions="16,8")
fill_weights(m)
m.to(dev).eval()
ms = [copy.deepcopy(m) for _ in range(4)]
x = mri_like(4, S, dev)
t = torch.full((4,), 500, device=dev, dtype=torch.int64)


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def graph_of(body):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g


with torch.no_grad():
    out4 = torch.empty_like(x)
    g4 = graph_of(lambda: m.forward_hip(x, t, out=out4))
    print("B=4 one stream, graph: %.3f ms" % timeit(g4.replay))
    for nw in (2, 4):
        nb = 4 // nw
        xs = [x[i * nb:(i + 1) * nb].contiguous() for i in range(nw)]
        ts = [t[i * nb:(i + 1) * nb].contiguous() for i in range(nw)]
        outs = [torch.empty_like(a) for a in xs]
        streams = [torch.cuda.Stream() for _ in range(nw - 1)]

        def body():
            cur = torch.cuda.current_stream()
            for i in range(nw):
                if i == 0:
                    ms[0].forward_hip(xs[0], ts[0], out=outs[0])
                else:
                    s = streams[i - 1]
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        ms[i].forward_hip(xs[i], ts[i], out=outs[i])
            for s in streams:
                cur.wait_stream(s)

        def serial():
            for i in range(nw):
                ms[i].forward_hip(xs[i], ts[i], out=outs[i])
        print("%d x B=%d serial, eager: %.3f ms" % (nw, nb, timeit(serial)))
        print("%d x B=%d on %d streams, eager: %.3f ms" % (nw, nb, nw, timeit(body)))
        gs = graph_of(serial)
        print("%d x B=%d serial, graph: %.3f ms" % (nw, nb, timeit(gs.replay)))
        gb = graph_of(body)
        print("%d x B=%d on %d streams, graph: %.3f ms" % (nw, nb, nw, timeit(gb.replay)))
        err = (torch.cat(outs) - out4).abs().max().item() / out4.abs().max().item()
        print("   max rel diff vs B=4:", err)
