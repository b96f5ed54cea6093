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
    dmol_head = args.x_like.split("_")[1] == "dmol" and getattr(args, "vae", vae) != "simple"
    if dmol_head:
        # vae.HVAE only builds the discretised-Gaussian head itself (the reference constructs -- but never raises -- its
        # NotImplementedError, vae.py:432-434, and swaps the head afterwards, SURVEY probe C.6): build with a placeholder
        from .dmol import DmolNet

        x_like = args.x_like
        args.x_like = x_like.split("_")[0] + "_dgauss"
        model = Model(args)
        args.x_like = x_like
        model.likelihood = DmolNet(args)
    else:
        model = Model(args)
    model.load_state_dict(ckpt["ema_model_state_dict" if which == "ema" else "model_state_dict"])
    if device is not None:
        model = model.to(device)
    return model, args


def resume_train_step(path: str, train_step) -> dict:
    """main.py:75-90: restore model / EMA weights and the optimiser + schedule state of ``train_step`` from ``path``.
    Returns the checkpoint's bookkeeping (epoch, step, best_loss).
    LR schedule: the warm-up position is derived from the optimiser's step count, i.e. a checkpoint taken inside the 100-step
    warm-up continues the ramp.  (The reference does not restore its scheduler at all: main.py:82-90 resets ``lr`` /
    ``initial_lr`` to ``args.lr`` under a constant LambdaLR, so a resumed reference run never re-enters warm-up.  Past step 100
    the two agree; inside the warm-up this harness deliberately keeps the ramp.)"""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    train_step.model.load_state_dict(ckpt["model_state_dict"])
    if train_step.ema_model is not None:
        train_step.ema_model.load_state_dict(ckpt["ema_model_state_dict"])
    if ckpt.get("optimizer_state_dict") is not None:
        train_step.load_state_dict(ckpt["optimizer_state_dict"])
        if "cgen" not in ckpt["optimizer_state_dict"] and ckpt.get("step") is not None:
            # a reference-made checkpoint: the ITERATION counter (beta warm-up, trainer.py:57) is ckpt["step"], which runs ahead of
            # the optimiser's step count whenever updates were skipped or accu_steps > 1
            train_step.it = int(ckpt["step"])
    train_step._mark_weights_written()
    return {k: ckpt.get(k) for k in ("epoch", "step", "best_loss")}


def save_checkpoint(path: str, model, ema_model, args, epoch: int = 0, step: int = 0, best_loss: float = float("inf"),
                    optimizer_state=None, scheduler_state=None, train_step=None) -> None:
    """trainer.py:154-165.  ``train_step`` (a ``train.TrainStep``) supplies ``optimizer_state_dict`` (AdamW layout: moments,
    step count) and ``scheduler_state_dict`` (LambdaLR warm-up position) so that a resume as in main.py:75-90 continues
    the moments, the LR warm-up and the EMA warm-up instead of restarting them."""
    if train_step is not None:
        optimizer_state = train_step.state_dict() if optimizer_state is None else optimizer_state
        scheduler_state = train_step.scheduler_state_dict() if scheduler_state is None else scheduler_state
    torch.save({"epoch": epoch, "step": step, "best_loss": float(best_loss), "model_state_dict": model.state_dict(),
                "ema_model_state_dict": ema_model.state_dict(), "optimizer_state_dict": optimizer_state,
                "scheduler_state_dict": scheduler_state, "hparams": dict(vars(args))}, path)
