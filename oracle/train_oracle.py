"""oracle/train_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Stock-PyTorch CPU restatement of the reference training-loop body (diffusion_training.py:99-107) over the
functional UNet oracle: p_loss (GaussianDiffusion.py:399-434, l2 branch, weights = 1) -> backward ->
clip_grad_norm_(1) -> AdamW(betas 0.9/0.999) -> EMA (UNet.py:423-427).  Checker for the HIP training step and the
"port" CPU baseline of `bench.py --config c3`; nothing under anoddpm_amd/ imports it.

Parity status: PINNED by tests/golden/train_*.npz -- two steps of the imported reference's own classes with the
random draws (t, forward noise, data) injected (tests/golden/make_golden.py:gen_training).
"""
import torch

from . import diffusion_oracle as do
from . import unet_oracle as uo


class TrainState:
    def __init__(self, sd, model_kw, lr=1e-4, weight_decay=0.0, ema_decay=0.9999):
        self.kw = dict(model_kw)
        self.params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        self.ema = {k: v.clone() for k, v in sd.items()}
        self.opt = torch.optim.AdamW(list(self.params.values()), lr=lr, weight_decay=weight_decay, betas=(0.9, 0.999))
        self.ema_decay = ema_decay
        self.tb = do.tables(do.beta_schedule(1000, "linear"))

    def loss(self, x0, t, noise):
        """p_loss with loss_type l2 and loss_weight 'none' (GaussianDiffusion.py:399-434)."""
        x_t = do.q_sample(self.tb, x0, t, noise)
        eps = uo.forward_autograd(self.params, x_t, t, **self.kw)
        per_image = (eps - noise).square().mean(dim=[1, 2, 3])
        return per_image.mean(), x_t, eps

    def step(self, x0, t, noise):
        """One pass of diffusion_training.py:99-107.  Returns (loss, pre-clip gradients, total norm)."""
        loss, x_t, eps = self.loss(x0, t, noise)
        self.opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in self.params.items()}
        norm = torch.nn.utils.clip_grad_norm_(list(self.params.values()), 1)
        self.opt.step()
        with torch.no_grad():
            for k, p in self.params.items():
                self.ema[k].mul_(self.ema_decay).add_(p.detach(), alpha=1 - self.ema_decay)
        return loss.detach(), grads, norm, x_t.detach(), eps.detach()
