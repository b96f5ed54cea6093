"""Hyper-parameter presets for the oracle (values only).

The numbers are the reference's preset *data* (hps.py:12-78) merged over the
argparse defaults the model reads (hps.py:94-254); the launch flags of
run_local.sh / run_slurm.sh:24-36 are folded into the named configs so a test
can ask for "the morphomnist HVAE as the authors ran it".
"""
from types import SimpleNamespace

_DEFAULTS = dict(
    hps="ukbb64", vae="hierarchical",
    enc_arch="64b1d2,32b1d2,16b1d2,8b1d8,1b2", dec_arch="1b2,8b2,16b2,32b2,64b2",
    widths=[16, 32, 48, 64, 128], bottleneck=4, z_dim=16, z_max_res=192, bias_max_res=64,
    input_res=64, input_channels=1, context_dim=4, cond_prior=False, q_correction=False,
    x_like="diag_dgauss", std_init=0.0, kl_free_bits=0.0, beta=1.0,
    lr=1e-3, wd=0.01, betas=[0.9, 0.9], lr_warmup_steps=100, ema_rate=0.999,
    grad_clip=350.0, grad_skip=500.0, accu_steps=1, bs=32,
)

_ARCH_192 = dict(
    enc_arch="192b1d2,96b3d2,48b7d2,24b11d2,12b7d2,6b3d6,1b2",
    dec_arch="1b2,6b4,12b8,24b12,48b8,96b4,192b2",
    widths=[32, 64, 96, 128, 160, 192, 512], input_res=192, z_dim=16,
)

PRESETS = {
    # hps.py:12-26 + run_local.sh (--cond_prior --concat_pa --context_dim 12)
    "morphomnist": dict(
        hps="morphomnist", input_res=32, input_channels=1, z_dim=16, wd=0.01,
        enc_arch="32b3d2,16b3d2,8b3d2,4b3d4,1b4", dec_arch="1b4,4b4,8b4,16b4,32b4",
        widths=[16, 32, 64, 128, 256], context_dim=12, cond_prior=True,
    ),
    # hps.py:29-42
    "cmnist": dict(
        hps="cmnist", input_res=32, input_channels=3, z_dim=16, wd=0.01,
        enc_arch="32b3d2,16b3d2,8b3d2,4b3d4,1b4", dec_arch="1b4,4b4,8b4,16b4,32b4",
        widths=[16, 32, 64, 128, 256], context_dim=20,
    ),
    # hps.py:45-55
    "ukbb64": dict(
        hps="ukbb64", input_res=64, z_dim=16, wd=0.1,
        enc_arch="64b3d2,32b31d2,16b15d2,8b7d2,4b3d4,1b2", dec_arch="1b2,4b4,8b8,16b16,32b32,64b4",
        widths=[32, 64, 128, 256, 512, 1024],
    ),
    # hps.py:58-65 + run_slurm.sh:24-36 (--context_dim 4 --beta 5 --z_max_res 96 --wd 0.05)
    "ukbb192": dict(hps="ukbb192", context_dim=4, beta=5.0, z_max_res=96, wd=0.05, **_ARCH_192),
    # hps.py:68-78 + run_slurm.sh:39-52 (commented block)
    "mimic192": dict(hps="mimic192", context_dim=6, beta=9.0, z_max_res=96, wd=0.05, bs=24, **_ARCH_192),
    # a 224^2 arch that the reference runs (7 is not a legal decoder res)
    "mimic224": dict(
        hps="mimic192", context_dim=6, beta=9.0, z_max_res=112, wd=0.05, bs=24, z_dim=16, input_res=224,
        enc_arch="224b1d2,112b3d2,56b7d2,28b11d2,14b7d2,8b3d8,1b2",
        dec_arch="1b2,8b4,14b8,28b12,56b8,112b4,224b2",
        widths=[32, 64, 96, 128, 160, 192, 512],
    ),
}


def make_hparams(name, **overrides):
    """Namespace with every field vae.HVAE.__init__ reads (vae.py:425-436)."""
    d = dict(_DEFAULTS)
    d.update(PRESETS[name])
    d.update(overrides)
    d["widths"] = list(d["widths"])
    return SimpleNamespace(**d)


def tiny_hparams(**overrides):
    """Small custom arch used by the golden fixtures."""
    d = dict(_DEFAULTS)
    d.update(
        hps="tiny", input_res=16, input_channels=1, z_dim=4, context_dim=3,
        enc_arch="16b1d2,8b1d8,1b1", dec_arch="1b1,8b2,16b1", widths=[8, 16, 32],
        z_max_res=16, bias_max_res=64,
    )
    d.update(overrides)
    d["widths"] = list(d["widths"])
    return SimpleNamespace(**d)
