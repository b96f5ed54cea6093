"""Parent-SCM counterfactuals without Pyro: the mechanisms of ``src/pgm/flow_pgm.py`` for the parents of
the image -- ``FlowPGM`` (UKBB: sex, mri_seq, age, brain_volume, ventricle_volume; flow_pgm.py:111-205) and
``MorphoMNISTPGM`` (thickness, intensity, digit; flow_pgm.py:314-385), ``ColourMNISTPGM`` (digit, colour: two categorical
roots; flow_pgm.py:451-530) and ``ChestPGM`` (race, sex, age -> finding; flow_pgm.py:533-710) -- restated in plain torch with
CLOSED-FORM abduction, so that ``DSCM.forward`` / ``dscm.counterfactual`` run on a box without pyro.

The reference gets abduction -> action -> prediction from Pyro effect handlers (``BasePGM.counterfactual``,
flow_pgm.py:71-108: trace the conditioned model, invert every TransformedDistribution's transforms at the observed value,
condition the reparameterised SCM on that noise, ``do`` the intervention, re-run).  For these DAGs that is:

    eps_k = f_k^{-1}(obs_k ; obs_pa(k))            abduction, per flow node, in closed form
    cf_k  = do_k                  if k is intervened on
          = obs_k                 if k is a root without a flow (sex, mri_seq, digit: "no exogenous noise available")
          = f_k(eps_k ; cf_pa(k)) otherwise          prediction, in topological order

with f = linear rational spline (pyro ``T.Spline(1, count_bins=4, order="linear")``), conditional affine
(layers.py:33-43: ``loc + exp(log_scale) * eps`` with (loc, log_scale) = DenseNN(context)), the [-1, 1] normalisation
``2 sigmoid(.) - 1``, and -- for discrete mechanisms -- the Gumbel-max posterior of layers.py:107-171.

Scalars only; there is no kernel here (a few floats per sample, host-side torch).  The parameter / module names follow
the reference's so that its ``state_dict`` keys load: ``load_reference_state_dict`` takes a reference PGM checkpoint, drops the
anticausal predictors the reference keeps on the same module (``encoder_*``: CNN / ResNet heads used by ``train_pgm.py`` and the
classifier guide, outside this path -- SURVEY 8f-3) and loads everything else strictly.

PARITY UNPINNED: pyro cannot be imported in the build image, so no reference-made vectors exist for this file.  It is pinned
by the invariants the mechanisms must satisfy (tests/test_pgm.py): f^{-1}(f(eps)) = eps, strict monotonicity, identity
outside the spline's bound, null intervention = observation, interventions move descendants only, Gumbel-max abduction
reproduces the observed class.  The spline follows Dolatabadi et al. 2020 ("Invertible generative modeling using linear
rational splines") as implemented by pyro 1.8 (``_monotonic_rational_spline``), restated from its published formulae.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# ----------------------------------------------------------------------------- pyro.nn.DenseNN, restated
class DenseNN(nn.Module):
    """input -> hidden_dims (nonlinearity between) -> sum(param_dims), split into one tensor per param dim."""

    def __init__(self, input_dim: int, hidden_dims: List[int], param_dims: List[int], nonlinearity: nn.Module):
        super().__init__()
        dims = [input_dim] + list(hidden_dims)
        layers = [nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]
        layers.append(nn.Linear(dims[-1], sum(param_dims)))
        self.layers = nn.ModuleList(layers)
        self.f = nonlinearity
        self.param_dims = list(param_dims)

    def forward(self, x: Tensor):
        h = x
        for lin in self.layers[:-1]:
            h = self.f(lin(h))
        h = self.layers[-1](h)
        if len(self.param_dims) == 1:
            return h
        return tuple(h.split(self.param_dims, dim=-1))


# ----------------------------------------------------------------------------- elementary bijections
class ConditionalAffine(nn.Module):
    """layers.py:33-43: y = loc(ctx) + exp(log_scale(ctx)) * x."""

    def __init__(self, context_nn: nn.Module):
        super().__init__()
        self.context_nn = context_nn

    def params(self, ctx: Tensor):
        loc, log_scale = self.context_nn(ctx)
        return loc, log_scale

    def forward(self, x: Tensor, ctx: Tensor) -> Tensor:
        loc, log_scale = self.params(ctx)
        return loc + log_scale.exp() * x

    def inv(self, y: Tensor, ctx: Tensor) -> Tensor:
        loc, log_scale = self.params(ctx)
        return (y - loc) / log_scale.exp()


def normalize_fwd(x: Tensor) -> Tensor:
    """ComposeTransform([SigmoidTransform(), AffineTransform(loc=-1, scale=2)]) (flow_pgm.py:327-330)."""
    return 2.0 * torch.sigmoid(x) - 1.0


def normalize_inv(y: Tensor) -> Tensor:
    p = ((y + 1.0) / 2.0).clamp(torch.finfo(y.dtype).tiny, 1.0 - torch.finfo(y.dtype).eps)  # torch's SigmoidTransform clamps alike
    return torch.log(p) - torch.log1p(-p)


class LinearSpline(nn.Module):
    """Element-wise monotone linear rational spline on [-bound, bound], identity outside (pyro T.Spline(order="linear")).

    Per bin k with knots (x_k, y_k), (x_{k+1}, y_{k+1}), derivatives d_k, d_{k+1} and an interior point lambda_k:
        w_a = 1,  w_b = sqrt(d_k / d_{k+1}),  w_c = (lambda w_a d_k + (1 - lambda) w_b d_{k+1}) / s_k,   s_k = (y_{k+1} - y_k) / (x_{k+1} - x_k)
        y_c = ((1 - lambda) w_a y_k + lambda w_b y_{k+1}) / ((1 - lambda) w_a + lambda w_b)
        theta = (x - x_k) / (x_{k+1} - x_k)
        y = (w_a y_k (lambda - theta) + w_c y_c theta) / (w_a (lambda - theta) + w_c theta)                  theta <= lambda
          = (w_c y_c (1 - theta) + w_b y_{k+1} (theta - lambda)) / (w_c (1 - theta) + w_b (theta - lambda))  theta >  lambda
    Both pieces are linear-fractional in theta, hence invertible in closed form."""

    def __init__(self, input_dim: int = 1, count_bins: int = 4, bound: float = 3.0):
        super().__init__()
        self.input_dim, self.count_bins, self.bound = input_dim, count_bins, bound
        self.min_bin_width = self.min_bin_height = self.min_derivative = 1e-3
        self.min_lambda = 0.025
        self.unnormalized_widths = nn.Parameter(torch.randn(input_dim, count_bins))
        self.unnormalized_heights = nn.Parameter(torch.randn(input_dim, count_bins))
        self.unnormalized_derivatives = nn.Parameter(torch.randn(input_dim, count_bins - 1))
        self.unnormalized_lambdas = nn.Parameter(torch.rand(input_dim, count_bins))

    def _knots(self):
        K, B = self.count_bins, self.bound
        w = self.min_bin_width + (1.0 - self.min_bin_width * K) * F.softmax(self.unnormalized_widths, dim=-1)
        h = self.min_bin_height + (1.0 - self.min_bin_height * K) * F.softmax(self.unnormalized_heights, dim=-1)
        d = self.min_derivative + F.softplus(self.unnormalized_derivatives)
        lam = (1.0 - 2.0 * self.min_lambda) * torch.sigmoid(self.unnormalized_lambdas) + self.min_lambda
        xk = F.pad(torch.cumsum(w, -1), (1, 0)) * 2 * B - B
        yk = F.pad(torch.cumsum(h, -1), (1, 0)) * 2 * B - B
        xk = torch.cat([xk[..., :1] * 0 - B, xk[..., 1:-1], xk[..., -1:] * 0 + B], -1)
        yk = torch.cat([yk[..., :1] * 0 - B, yk[..., 1:-1], yk[..., -1:] * 0 + B], -1)
        one = torch.ones_like(d[..., :1])
        d = torch.cat([one, d, one], -1)  # linear tails: boundary derivatives 1
        return xk, yk, d, lam

    def _bin(self, v: Tensor, knots: Tensor) -> Tensor:
        # v [..., D], knots [D, K+1] -> bin index in [0, K-1]
        idx = (v.unsqueeze(-1) >= knots[..., 1:-1]).sum(-1)
        return idx.clamp(0, self.count_bins - 1)

    def _coeffs(self, idx: Tensor):
        xk, yk, d, lam = self._knots()

        def take(t, off=0):
            ex = t.expand(idx.shape[:-1] + t.shape) if t.dim() == 2 else t
            return torch.gather(ex, -1, (idx + off).unsqueeze(-1)).squeeze(-1)

        x0, x1, y0, y1 = take(xk), take(xk, 1), take(yk), take(yk, 1)
        d0, d1, lm = take(d), take(d, 1), take(lam)
        wa = torch.ones_like(d0)
        wb = torch.sqrt(d0 / d1) * wa
        s = (y1 - y0) / (x1 - x0)
        wc = (lm * wa * d0 + (1.0 - lm) * wb * d1) / s
        yc = ((1.0 - lm) * wa * y0 + lm * wb * y1) / ((1.0 - lm) * wa + lm * wb)
        return x0, x1, y0, y1, lm, wa, wb, wc, yc

    def forward(self, x: Tensor) -> Tensor:
        inside = (x > -self.bound) & (x < self.bound)
        xc = x.clamp(-self.bound, self.bound)
        xk = self._knots()[0]
        idx = self._bin(xc, xk)
        x0, x1, y0, y1, lm, wa, wb, wc, yc = self._coeffs(idx)
        th = (xc - x0) / (x1 - x0)
        left = (wa * y0 * (lm - th) + wc * yc * th) / (wa * (lm - th) + wc * th)
        right = (wc * yc * (1.0 - th) + wb * y1 * (th - lm)) / (wc * (1.0 - th) + wb * (th - lm))
        y = torch.where(th <= lm, left, right)
        return torch.where(inside, y, x)

    def inv(self, y: Tensor) -> Tensor:
        inside = (y > -self.bound) & (y < self.bound)
        ycl = y.clamp(-self.bound, self.bound)
        yk = self._knots()[1]
        idx = self._bin(ycl, yk)
        x0, x1, y0, y1, lm, wa, wb, wc, yc = self._coeffs(idx)
        # solve the linear-fractional pieces for theta
        th_l = lm * wa * (y0 - ycl) / (wa * (y0 - ycl) + wc * (ycl - yc))
        th_r = (wc * (yc - ycl) + lm * wb * (ycl - y1)) / (wc * (yc - ycl) + wb * (ycl - y1))
        th = torch.where(ycl <= yc, th_l, th_r)
        x = th * (x1 - x0) + x0
        return torch.where(inside, x, y)


# ----------------------------------------------------------------------------- Gumbel-max mechanism (layers.py:107-171)
def gumbel_max_forward(gumbels: Tensor, logits: Tensor) -> Tensor:
    """ArgMaxGumbelMax._call: the class is argmax_k (g_k + logit_k)."""
    return (gumbels + logits).argmax(-1, keepdim=True)


def gumbel_max_abduct(k: Tensor, logits: Tensor, generator: Optional[torch.Generator] = None) -> Tensor:
    """ArgMaxGumbelMax.inv (layers.py:139-160), restated as is: a fresh standard Gumbel g_k for the observed class, whose noise
    becomes g_k - logit_k; every other class gets a Gumbel truncated below that value.  NOTE (reference behaviour, kept): the
    truncation level is g_k - logit_k while the winner's perturbed logit is g_k, so with log-probabilities (logit_k < 0) another
    class can still exceed the winner -- the reference therefore pins ``finding`` to its observed value unless it or its
    parent is intervened on (flow_pgm.py:96-105).  `gumbel_max_abduct_exact` is the exact posterior."""
    u = torch.rand(logits.shape, dtype=logits.dtype, device=logits.device, generator=generator)
    gumbels = -(-(u.log())).log()
    mask = F.one_hot(k.squeeze(-1).to(torch.int64), num_classes=logits.shape[-1]).to(logits.dtype)
    top = (mask * gumbels).sum(-1, keepdim=True) - (mask * logits).sum(-1, keepdim=True)
    other = 1.0 - mask
    g = gumbels + logits
    return -torch.log(other * torch.exp(-g) + torch.exp(-top)) - other * logits


def gumbel_max_abduct_exact(k: Tensor, logits: Tensor, generator: Optional[torch.Generator] = None) -> Tensor:
    """Exact Gumbel-max posterior (Maddison et al. 2014, top-down): the maximum Z ~ Gumbel(logsumexp(logits)) is attained by
    class k; every other perturbed logit is a Gumbel(logit_i) truncated at Z.  Returns the NOISE (perturbed logit - logit), so
    gumbel_max_forward(result, logits) == k always."""
    u = torch.rand(logits.shape, dtype=logits.dtype, device=logits.device, generator=generator)
    g = -(-(u.log())).log() + logits                      # independent Gumbel(logit_i)
    mask = F.one_hot(k.squeeze(-1).to(torch.int64), num_classes=logits.shape[-1]).to(logits.dtype)
    z = torch.logsumexp(logits, -1, keepdim=True) + (mask * (g - logits)).sum(-1, keepdim=True)   # the maximum (fresh Gumbel)
    trunc = -torch.log(torch.exp(-g) + torch.exp(-z))    # Gumbel(logit_i) truncated at z
    y = mask * z + (1.0 - mask) * trunc
    return y - logits


# ----------------------------------------------------------------------------- the two parent SCMs
class _BasePGM(nn.Module):
    """Abduction / action / prediction over a small DAG of flow mechanisms (BasePGM.counterfactual, flow_pgm.py:71-108)."""
    variables: Dict[str, str] = {}
    order: List[str] = []

    def _abduct(self, obs: Dict[str, Tensor]) -> Dict[str, Tensor]:
        raise NotImplementedError

    def _predict(self, k: str, eps: Tensor, val: Dict[str, Tensor]) -> Tensor:
        raise NotImplementedError

    @torch.no_grad()
    def infer_exogeneous(self, obs: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {k + "_base": v for k, v in self._abduct(obs).items()}

    def counterfactual(self, obs: Dict[str, Tensor], intervention: Dict[str, Tensor], num_particles: int = 1,
                       detach: bool = True) -> Dict[str, Tensor]:
        assert set(obs.keys()) == set(self.variables.keys()), (sorted(obs), sorted(self.variables))
        avg = {k: torch.zeros_like(obs[k]) for k in obs}
        for _ in range(num_particles):
            eps = self._abduct(obs)
            if detach:
                eps = {k: v.detach() for k, v in eps.items()}
            val: Dict[str, Tensor] = {}
            for k in self.order:
                if k in intervention:
                    val[k] = intervention[k]
                elif k not in eps:  # root without a flow: no exogenous noise available -> the observed value
                    val[k] = obs[k]
                else:
                    val[k] = self._predict(k, eps[k], val)
            if getattr(self, "discrete_variables", None) and "age" not in intervention and "finding" not in intervention:
                # flow_pgm.py:96-105: abduction of a discrete mechanism is stochastic (Gumbel-max posterior), so "finding" keeps
                # its observed value unless it or its parent is intervened on
                val["finding"] = obs["finding"]
            for k in obs:
                avg[k] = avg[k] + val[k] / num_particles
        return avg

    def load_reference_state_dict(self, sd: Dict[str, Tensor]):
        """Load a reference PGM ``state_dict`` (train_cf.py:340-347 loads it into the pyro module).  The reference keeps its
        anticausal predictors on the same module (``encoder_s/m/a/b/v``, ``encoder_t/i/y``, ``encoder_y/c``, ``encoder_s/r/f/a``);
        they are not part of the counterfactual path and have no counterpart here: those keys are dropped, every other key must
        match exactly."""
        kept = {k: v for k, v in sd.items() if not k.split(".")[0].startswith("encoder_")}
        dropped = sorted(set(sd) - set(kept))
        self.load_state_dict(kept, strict=True)
        return dropped


class FlowPGM(_BasePGM):
    """UKBB parents (flow_pgm.py:111-205): s -> b <- a -> v <- b; m a root.  Variables are [B, 1] tensors, continuous ones
    in the [-1, 1] normalisation the PGM was trained with."""
    variables = {"sex": "binary", "mri_seq": "binary", "age": "continuous", "brain_volume": "continuous", "ventricle_volume": "continuous"}
    order = ["sex", "mri_seq", "age", "brain_volume", "ventricle_volume"]

    def __init__(self, args):
        super().__init__()
        self.s_logit = nn.Parameter(torch.zeros(1))
        self.m_logit = nn.Parameter(torch.zeros(1))
        for k in ("a", "b", "v"):
            self.register_buffer(f"{k}_base_loc", torch.zeros(1))
            self.register_buffer(f"{k}_base_scale", torch.ones(1))
        widths = list(getattr(args, "widths", [32, 32]))
        self.age_module = nn.ModuleList([LinearSpline(1, count_bins=4)])
        self.bvol_flow = ConditionalAffine(DenseNN(2, widths, [1, 1], nn.LeakyReLU(0.1)))
        self.vvol_flow = ConditionalAffine(DenseNN(2, widths, [1, 1], nn.LeakyReLU(0.1)))

    def _abduct(self, obs):
        return {"age": self.age_module[0].inv(obs["age"]),
                "brain_volume": self.bvol_flow.inv(obs["brain_volume"], torch.cat([obs["sex"], obs["age"]], 1)),
                "ventricle_volume": self.vvol_flow.inv(obs["ventricle_volume"], torch.cat([obs["brain_volume"], obs["age"]], 1))}

    def _predict(self, k, eps, val):
        if k == "age":
            return self.age_module[0](eps)
        if k == "brain_volume":
            return self.bvol_flow(eps, torch.cat([val["sex"], val["age"]], 1))
        return self.vvol_flow(eps, torch.cat([val["brain_volume"], val["age"]], 1))

    @torch.no_grad()
    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
        dev = self.s_logit.device
        r = lambda: torch.rand(n, 1, generator=generator).to(dev)
        z = lambda: torch.randn(n, 1, generator=generator).to(dev)
        val = {"sex": (r() < torch.sigmoid(self.s_logit)).float(), "mri_seq": (r() < torch.sigmoid(self.m_logit)).float()}
        for k in ("age", "brain_volume", "ventricle_volume"):
            val[k] = self._predict(k, z(), val)
        return val


class MorphoMNISTPGM(_BasePGM):
    """Morpho-MNIST parents (flow_pgm.py:314-385): thickness -> intensity; digit a categorical root (one-hot [B, 10])."""
    variables = {"thickness": "continuous", "intensity": "continuous", "digit": "categorical"}
    order = ["digit", "thickness", "intensity"]

    def __init__(self, args):
        super().__init__()
        self.digit_logits = nn.Parameter(torch.zeros(1, 10))
        for k in ("t", "i"):
            self.register_buffer(f"{k}_base_loc", torch.zeros(1))
            self.register_buffer(f"{k}_base_scale", torch.ones(1))
        widths = list(getattr(args, "widths", [32, 32]))
        self.thickness_module = nn.ModuleList([LinearSpline(1, count_bins=4)])
        self.context_nn = ConditionalAffine(DenseNN(1, widths, [1, 1], nn.GELU()))

    def _abduct(self, obs):
        return {"thickness": self.thickness_module[0].inv(normalize_inv(obs["thickness"])),
                "intensity": self.context_nn.inv(normalize_inv(obs["intensity"]), obs["thickness"])}

    def _predict(self, k, eps, val):
        if k == "thickness":
            return normalize_fwd(self.thickness_module[0](eps))
        return normalize_fwd(self.context_nn(eps, val["thickness"]))

    @torch.no_grad()
    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
        dev = self.digit_logits.device
        probs = F.softmax(self.digit_logits, -1).expand(n, -1).cpu()
        val = {"digit": F.one_hot(torch.multinomial(probs, 1, generator=generator).squeeze(-1), 10).float().to(dev)}
        for k in ("thickness", "intensity"):
            val[k] = self._predict(k, torch.randn(n, 1, generator=generator).to(dev), val)
        return val


class ColourMNISTPGM(_BasePGM):
    """Colour-MNIST parents (flow_pgm.py:451-530): digit and colour, two independent one-hot categorical ROOTS ([B, 10] each)
    with learned prior logits.  Neither has a flow, hence no exogenous noise: a counterfactual keeps the observed value of
    whatever is not intervened on (BasePGM.counterfactual, flow_pgm.py:84-88)."""
    variables = {"digit": "categorical", "colour": "categorical"}
    order = ["digit", "colour"]

    def __init__(self, args=None):
        super().__init__()
        self.digit_logits = nn.Parameter(torch.zeros(1, 10))   # uniform prior
        self.colour_logits = nn.Parameter(torch.zeros(1, 10))  # uniform prior

    def _abduct(self, obs):
        return {}

    @torch.no_grad()
    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
        dev = self.digit_logits.device
        draw = lambda logits: F.one_hot(torch.multinomial(F.softmax(logits, -1).expand(n, -1).cpu(), 1, generator=generator).squeeze(-1), 10).float().to(dev)
        return {"digit": draw(self.digit_logits), "colour": draw(self.colour_logits)}


class ChestPGM(_BasePGM):
    """MIMIC-CXR parents (flow_pgm.py:533-710): race (one-hot [B, 3]) and sex ([B, 1]) are roots without a flow; age ~ Normal ->
    ``T.Spline(1)`` (pyro defaults: 8 bins, bound 3, linear order); finding | age is a Gumbel-max mechanism whose two logits are
    ``DenseNN(1, [8, 16], [2], Sigmoid)(age)`` (layers.py:107-171, 174-195).  Abduction of `finding` is the stochastic
    posterior of layers.py:139-160 ([B, 2] Gumbel noise, `gumbel_max_abduct`); the reference therefore keeps the observed
    finding unless `age` or `finding` is intervened on (flow_pgm.py:96-105), and so does `_BasePGM.counterfactual`."""
    variables = {"race": "categorical", "sex": "binary", "finding": "binary", "age": "continuous"}
    discrete_variables = {"finding": "binary"}
    order = ["sex", "age", "race", "finding"]

    def __init__(self, args=None, exact_gumbel_posterior: bool = False):
        super().__init__()
        for k in ("a", "f"):
            self.register_buffer(f"{k}_base_loc", torch.zeros(1))
            self.register_buffer(f"{k}_base_scale", torch.ones(1))
        self.age_flow_components = nn.ModuleList([LinearSpline(1, count_bins=8, bound=3.0)])
        self.finding_transform_GumbelMax = nn.Module()
        self.finding_transform_GumbelMax.context_nn = DenseNN(1, [8, 16], [2], nn.Sigmoid())
        self.sex_logit = nn.Parameter(math.log(1 / 2) * torch.ones(1))
        self.race_logits = nn.Parameter(math.log(1 / 3) * torch.ones(1, 3))
        self.exact_gumbel_posterior = exact_gumbel_posterior
        self.generator: Optional[torch.Generator] = None  # set for reproducible Gumbel abduction (tests)

    def finding_logits(self, age: Tensor) -> Tensor:
        return self.finding_transform_GumbelMax.context_nn(age)

    def _abduct(self, obs):
        post = gumbel_max_abduct_exact if self.exact_gumbel_posterior else gumbel_max_abduct
        return {"age": self.age_flow_components[0].inv(obs["age"]),
                "finding": post(obs["finding"], self.finding_logits(obs["age"]), self.generator)}

    def _predict(self, k, eps, val):
        if k == "age":
            return self.age_flow_components[0](eps)
        return gumbel_max_forward(eps, self.finding_logits(val["age"])).to(eps.dtype)

    @torch.no_grad()
    def sample(self, n: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
        dev = self.sex_logit.device
        r = lambda *shape: torch.rand(*shape, generator=generator).to(dev)
        val = {"sex": (r(n, 1) < torch.sigmoid(self.sex_logit)).float()}
        val["age"] = self._predict("age", torch.randn(n, 1, generator=generator).to(dev), val)
        val["race"] = F.one_hot(torch.multinomial(F.softmax(self.race_logits, -1).expand(n, -1).cpu(), 1, generator=generator).squeeze(-1), 3).float().to(dev)
        gum = -(-(r(n, 2).log())).log()
        val["finding"] = self._predict("finding", gum, val)
        return val
