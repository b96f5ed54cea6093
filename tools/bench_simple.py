#!/usr/bin/env python3
"""Config 1 (BASELINE.json configs[0]: Morpho-MNIST 32x32 simple_vae, batch 32) through the fused train step on the GPU:
images/s at batch 32 and 256, f32 (the model's compute type), hipGraph replay.  usage: python tools/bench_simple.py [x_like]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from causal_gen_amd import simple_vae
from causal_gen_amd.hps import Hparams
from causal_gen_amd.train import TrainStep

x_like = sys.argv[1] if len(sys.argv) > 1 else "diag_dgauss"
C = 3 if x_like.endswith("dmol") else 1
hpd = dict(hps="morphomnist", input_res=32, input_channels=C, z_dim=16, context_dim=12, cond_prior=True, widths=[16, 32, 64, 128, 256],
           x_like=x_like, std_init=0.0, lr=1e-3, betas=(0.9, 0.9), wd=0.01, grad_clip=350.0, grad_skip=5000.0, ema_rate=0.999,
           lr_warmup_steps=100, beta=1.0, accu_steps=1, kl_free_bits=0.0)
for B in (32, 256):
    torch.manual_seed(0)
    m = simple_vae.VAE(Hparams(**hpd)).cuda().train()
    ts = TrainStep(m, SimpleNamespace(**hpd), ema=True, use_graph=True)
    x = ((torch.randint(0, 256, (B, C, 32, 32)).float() - 127.5) / 127.5).cuda()
    pa = torch.randn(B, 12).cuda()
    for _ in range(30):  # every conditioning-dropout outcome gets its graph
        o = ts.step(x, pa)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        o = ts.step(x, pa)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("simple_vae %s batch %d: %.0f images/s, %.3f ms/step, elbo %.4f" % (x_like, B, B / dt, dt * 1e3, float(o[0])), flush=True)
