"""Experiment (round 4): the batch of four as TWO free-running half-batch chains on two hardware queues, started a fraction of a
step apart, so that one half's latency-bound small-map section runs beside the other half's chip-filling large-map section
(round 3 joined the two streams after every forward: in phase by construction, 10.4 vs 9.95 ms).
  python tools/exp_offset.py        -> ms per four images for offsets 0 / 0.25 / 0.5 of a half-batch step, vs one B = 4 graph"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from UNet import UNetModel  # noqa: E402
from bench import fill_weights, mri_like  # noqa: E402

dev = torch.device("cuda:0")
m = UNetModel(256, 128, n_heads=2, attention_resolutions="16,8")
fill_weights(m)
m.to(dev).eval()
m2 = copy.deepcopy(m)
x = mri_like(4, 256, dev)
t = torch.full((4,), 500, device=dev, dtype=torch.int64)
xs = [x[:2].contiguous(), x[2:].contiguous()]
ts = [t[:2].contiguous(), t[2:].contiguous()]
outs = [torch.empty_like(a) for a in xs]
out4 = torch.empty_like(x)
N = 12


def graph_on(stream, fn):
    with torch.cuda.stream(stream):
        fn()
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    torch.cuda.synchronize()
    return g


with torch.no_grad():
    s0 = torch.cuda.Stream()
    g4 = graph_on(s0, lambda: m.forward_hip(x, t, out=out4))

    def run4():
        with torch.cuda.stream(s0):
            for _ in range(N):
                g4.replay()
    run4(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run4(); torch.cuda.synchronize()
    t4 = (time.perf_counter() - t0) / N * 1e3
    print("B=4 one graph: %.3f ms per 4 images" % t4, flush=True)
    streams = [torch.cuda.Stream() for _ in range(6)]
    gA = graph_on(streams[0], lambda: m.forward_hip(xs[0], ts[0], out=outs[0]))

    def runA():
        with torch.cuda.stream(streams[0]):
            for _ in range(N):
                gA.replay()
    runA(); torch.cuda.synchronize()
    t0 = time.perf_counter(); runA(); torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / N * 1e3
    print("B=2 one graph alone: %.3f ms per 2 images" % t2, flush=True)
    for k in range(1, 6):
        sB = streams[k]
        gB = graph_on(sB, lambda: m2.forward_hip(xs[1], ts[1], out=outs[1]))
        for frac in (0.0, 0.25, 0.5, 0.75):
            def both():
                with torch.cuda.stream(streams[0]):
                    gA.replay()
                if frac:
                    time.sleep(frac * t2 / 1e3)
                for i in range(N):
                    with torch.cuda.stream(sB):
                        gB.replay()
                    if i + 1 < N:
                        with torch.cuda.stream(streams[0]):
                            gA.replay()
            both(); torch.cuda.synchronize()
            t0 = time.perf_counter(); both(); torch.cuda.synchronize()
            tt = (time.perf_counter() - t0) / N * 1e3
            print("stream #%d offset %.2f: %.3f ms per 4 images (serial would be %.3f)" % (k, frac, tt, 2 * t2), flush=True)
    err = (torch.cat(outs) - out4).abs().max().item() / out4.abs().max().item()
    print("max rel diff vs B=4:", err)
