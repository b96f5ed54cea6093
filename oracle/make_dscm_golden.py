"""tests/golden/dscm_*.pt: outputs of the REFERENCE's own ``DSCM.forward`` (src/pgm/dscm.py:15-95), generated in the build
container:

    python3 -B oracle/make_dscm_golden.py

``dscm.py`` imports pyro (via ``layers``), torchvision (via ``datasets``), imageio (via ``utils``) and seaborn (via
``utils_pgm``), none of which exist here and none of which its ``forward`` touches: they are stubbed in ``sys.modules``
(empty modules whose attributes are placeholder classes) so that the REAL ``dscm.DSCM``, ``dscm.vae_preprocess`` /
``ukbb_preprocess``, ``utils_pgm.check_nan`` and ``datasets.get_attr_max_min`` run.  The parent SCM, predictor and Pyro ELBO are
the duck-typed stand-ins of oracle/dscm_stubs.py; ``Tensor.cuda`` is made a no-op (dscm.py:131 hard-codes it).  Fixtures are
data: inputs, injected noise, the reference's outputs and gradients."""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REF, os.path.join(REF, "pgm"), ROOT]

import torch  # noqa: E402


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = type(k, (), {})
        setattr(self, k, v)
        return v


for name in ("pyro", "seaborn", "torchvision", "torchvision.transforms", "imageio"):  # (not installed here)
    sys.modules[name] = _Stub(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
layers = _Stub("layers")  # (pgm/layers.py is pyro code; dscm.py only names TraceStorage_ELBO in a type annotation)
sys.modules["layers"] = layers

import vae as ref_vae  # noqa: E402  (reference)
import dscm as ref_dscm  # noqa: E402  (reference: src/pgm/dscm.py)
from hps import Hparams  # noqa: E402  (reference)

from oracle import dscm_stubs as S  # noqa: E402
from oracle import hparams as ohp  # noqa: E402
from oracle.make_golden import EpsTap, randomise  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # dscm.py:131
OUT = os.path.join(ROOT, "tests", "golden")


def case(tag, B=3, seed=0):
    over, dataset, parents_x, do_keys, particles, t_abduct = S.CASES[tag]
    hp = ohp.tiny_hparams(**over)
    gen = torch.Generator().manual_seed(4000 + seed)
    torch.manual_seed(seed)
    a = Hparams()
    a.update(dict(vars(hp)))
    a.update(dict(dataset=dataset, parents_x=parents_x, **S.CONSTANTS))
    m = ref_vae.HVAE(a)
    randomise(m, gen)
    m.eval()
    obs, do = S.make_obs(tag, hp, B, gen)
    w = torch.randn(obs["x"].shape, generator=gen) * 0.3
    model = ref_dscm.DSCM(a, S.StubPGM(), S.StubPredictor(), m)
    torch.manual_seed(31 + seed)
    with EpsTap() as tap:
        out = model({k: v.clone() for k, v in obs.items()}, {k: v.clone() for k, v in do.items()}, S.StubELBO(w), cf_particles=particles,
                    t_abduct=t_abduct)
    out["loss"].sum().backward()
    fx = dict(tag=tag, hp=dict(vars(hp)), dataset=dataset, parents_x=parents_x, particles=particles, t_abduct=t_abduct, constants=dict(S.CONSTANTS),
              obs=obs, do=do, w=w, eps=tap.eps, state_dict={k: v.detach().clone() for k, v in m.state_dict().items()},
              out={k: out[k].detach().clone() for k in ("elbo", "nll", "kl", "loss", "aux_loss")},
              cf_x=out["cfs"]["x"].detach().clone(), cf_parents={k: v.detach().clone() for k, v in out["cfs"].items() if k != "x"},
              var_cf_x=None if out["var_cf_x"] is None else out["var_cf_x"].detach().clone(),
              lmbda_grad=model.lmbda.grad.clone(), grads={n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
              vae_parents=ref_dscm.vae_preprocess(a, {k: v.clone() for k, v in obs.items() if k != "x"})[:, :, 0, 0].clone())
    path = os.path.join(OUT, "dscm_%s.pt" % tag)
    torch.save(fx, path)
    print("%s: %d KiB  loss %.6f aux %.6f elbo %.6f  %d draws, %d gradients" % (path, os.path.getsize(path) // 1024, float(out["loss"]), float(out["aux_loss"]),
                                                                                 float(out["elbo"]), len(tap.eps), len(fx["grads"])))


if __name__ == "__main__":
    for i, tag in enumerate(S.CASES):
        case(tag, seed=i)
