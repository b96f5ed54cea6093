"""Checkpoint wire format of the reference: ``trainer.py:154-165`` writes ``checkpoint.pt`` with
``model_state_dict`` / ``ema_model_state_dict`` / ``optimizer_state_dict`` / ``scheduler_state_dict`` / ``hparams``;
``train_cf.py:357-364`` loads the EMA weights into ``HVAE(Hparams(**hparams))`` after the ``free_bits`` ->
``kl_free_bits`` rename and a ``cond_prior`` default.  The module tree here has the reference's key names, so the
state dicts load unchanged; this file only reproduces that loader / saver around them."""
from typing import Optional, Tuple

import torch

from .hps import Hparams


def hparams_from_checkpoint(ckpt) -> Hparams:
    """train_cf.py:359-362."""
    args = Hparams()
    args.update(dict(ckpt["hparams"]))
    if not hasattr(args, "cond_prior"):  # backwards compatibility in the reference
        args.cond_prior = False
    if hasattr(args, "free_bits") and not hasattr(args, "kl_free_bits"):
        args.kl_free_bits = args.free_bits
    return args


def load_checkpoint(path: str, which: str = "ema", device: Optional[str] = "cuda", vae: str = "hierarchical") -> Tuple[torch.nn.Module, Hparams]:
    """Build the model a reference ``checkpoint.pt`` describes and load its weights (``which``: "ema" | "model")."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    args = hparams_from_checkpoint(ckpt)
    if getattr(args, "vae", vae) == "simple":
        from .simple_vae import VAE as Model
    else:
        from .vae import HVAE as Model
    model = Model(args)
    if args.x_like.split("_")[1] == "dmol" and getattr(args, "vae", vae) != "simple":
        from .dmol import DmolNet

        model.likelihood = DmolNet(args)
    model.load_state_dict(ckpt["ema_model_state_dict" if which == "ema" else "model_state_dict"])
    if device is not None:
        model = model.to(device)
    return model, args


def save_checkpoint(path: str, model, ema_model, args, epoch: int = 0, step: int = 0, best_loss: float = float("inf"),
                    optimizer_state=None, scheduler_state=None) -> None:
    """trainer.py:154-165 (``optimizer_state`` / ``scheduler_state`` are whatever the caller's step harness keeps)."""
    torch.save({"epoch": epoch, "step": step, "best_loss": float(best_loss), "model_state_dict": model.state_dict(),
                "ema_model_state_dict": ema_model.state_dict(), "optimizer_state_dict": optimizer_state,
                "scheduler_state_dict": scheduler_state, "hparams": dict(vars(args))}, path)
