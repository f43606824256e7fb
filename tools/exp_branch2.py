"""Which side streams actually overlap with the capture stream?  2 x B=2 forwards, side stream = k-th created stream."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from UNet import UNetModel
from bench import fill_weights, mri_like
dev = torch.device("cuda:0")
m = UNetModel(256, 128, n_heads=2, attention_resolutions="16,8")
fill_weights(m); m.to(dev).eval()
m2 = copy.deepcopy(m)
x = mri_like(4, 256, dev); t = torch.full((4,), 500, device=dev, dtype=torch.int64)
xs = [x[:2].contiguous(), x[2:].contiguous()]; ts = [t[:2].contiguous(), t[2:].contiguous()]
outs = [torch.empty_like(a) for a in xs]

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    m.forward_hip(xs[0], ts[0], out=outs[0]); m2.forward_hip(xs[1], ts[1], out=outs[1]); torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(10)]
    main = torch.cuda.Stream()
    for k, s in enumerate(streams):
        def body():
            with torch.cuda.stream(main):
                s.wait_stream(main)
                m.forward_hip(xs[0], ts[0], out=outs[0])
                with torch.cuda.stream(s):
                    m2.forward_hip(xs[1], ts[1], out=outs[1])
                main.wait_stream(s)
        print("eager side stream #%d: %.3f ms" % (k, timeit(body)), flush=True)
    # raw HIP streams with explicit flags via ctypes? use priorities
    for pr in (-1, 0):
        s = torch.cuda.Stream(priority=pr)
        def body():
            with torch.cuda.stream(main):
                s.wait_stream(main)
                m.forward_hip(xs[0], ts[0], out=outs[0])
                with torch.cuda.stream(s):
                    m2.forward_hip(xs[1], ts[1], out=outs[1])
                main.wait_stream(s)
        print("eager side stream priority %d: %.3f ms" % (pr, timeit(body)), flush=True)
