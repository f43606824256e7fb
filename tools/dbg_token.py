import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import GaussianDiffusion as GD
from UNet import UNetModel
from anoddpm_amd import unet as U
orig = U._Plan.refresh_weights
def patched(self):
    params = list(self.model.parameters())
    token = (self.model._weights_epoch,) + tuple((p.data_ptr(), p._version) for p in params)
    same = token == self.token
    if not same and self.token is not None:
        diffs = [(i, a, b) for i, (a, b) in enumerate(zip(token, self.token)) if a != b]
        print("TOKEN CHANGED", len(diffs), diffs[:3], "capturing:", torch.cuda.is_current_stream_capturing())
    return orig(self)
U._Plan.refresh_weights = patched
m = UNetModel(64, 32, n_heads=2, attention_resolutions="16,8").to("cuda:0").eval()
d = GD.GaussianDiffusionModel([64, 64], GD.get_beta_schedule(1000, "linear"), noise="simplex")
x = torch.rand(2, 1, 64, 64, device="cuda:0")
ch = GD.ReverseChain(d, m, x, 6, "simplex", use_graph=True)
for i in range(4):
    print("step", i, "state", ch._graph_state)
    ch.step()
print("ok")
