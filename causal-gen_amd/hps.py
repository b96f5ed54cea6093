"""Hyper-parameter bag and presets: the fields ``HVAE``/``DSCM``/the train step read (hps.py:3-254 restated as data)."""


class Hparams:
    def __init__(self, **kw):
        self.update(kw)

    def update(self, d):
        for k, v in d.items():
            setattr(self, k, v)


_DEFAULTS = dict(
    hps="ukbb64", vae="hierarchical", seed=7, bs=32, lr=1e-3, lr_warmup_steps=100, wd=0.01, betas=[0.9, 0.9],
    ema_rate=0.999, input_res=64, input_channels=1, grad_clip=350.0, grad_skip=500.0, accu_steps=1, beta=1.0,
    beta_warmup_steps=0, kl_free_bits=0.0, enc_arch="64b1d2,32b1d2,16b1d2,8b1d8,1b2", dec_arch="1b2,8b2,16b2,32b2,64b2",
    cond_prior=False, widths=[16, 32, 48, 64, 128], bottleneck=4, z_dim=16, z_max_res=192, bias_max_res=64,
    x_like="diag_dgauss", std_init=0.0, parents_x=["mri_seq", "brain_volume", "ventricle_volume", "sex"], concat_pa=False,
    context_dim=4, context_norm="log_standard", q_correction=False, dataset="",
)
_A192 = dict(enc_arch="192b1d2,96b3d2,48b7d2,24b11d2,12b7d2,6b3d6,1b2", dec_arch="1b2,6b4,12b8,24b12,48b8,96b4,192b2",
             widths=[32, 64, 96, 128, 160, 192, 512], input_res=192, z_dim=16)
_MNIST = dict(enc_arch="32b3d2,16b3d2,8b3d2,4b3d4,1b4", dec_arch="1b4,4b4,8b4,16b4,32b4", widths=[16, 32, 64, 128, 256],
              input_res=32, z_dim=16, wd=0.01)

HPARAMS_REGISTRY = {
    "morphomnist": dict(hps="morphomnist", parents_x=["thickness", "intensity", "digit"], concat_pa=True,
                        context_norm="[-1,1]", context_dim=12, cond_prior=True, **_MNIST),
    "cmnist": dict(hps="cmnist", input_channels=3, parents_x=["digit", "colour"], context_dim=20, **_MNIST),
    "ukbb192": dict(hps="ukbb192", wd=0.05, beta=5.0, z_max_res=96, context_dim=4, concat_pa=True, dataset="ukbb", **_A192),
    "mimic192": dict(hps="mimic192", wd=0.05, beta=9.0, z_max_res=96, context_dim=6, bs=24,
                     parents_x=["age", "race", "sex", "finding"], **_A192),
    "mimic224": dict(hps="mimic192", wd=0.05, beta=9.0, z_max_res=112, context_dim=6, bs=24, input_res=224, z_dim=16,
                     parents_x=["age", "race", "sex", "finding"],
                     enc_arch="224b1d2,112b3d2,56b7d2,28b11d2,14b7d2,8b3d8,1b2", dec_arch="1b2,8b4,14b8,28b12,56b8,112b4,224b2",
                     widths=[32, 64, 96, 128, 160, 192, 512]),
}


def setup_hparams(name, **overrides):
    d = dict(_DEFAULTS)
    d.update(HPARAMS_REGISTRY[name])
    d.update(overrides)
    d["widths"] = list(d["widths"])
    return Hparams(**d)
