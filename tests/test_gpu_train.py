"""Fused train step (forward, hand-written backward, device-side clip/skip, AdamW+EMA) against the oracle's
restatement of trainer.py:54-87; hipGraph replay against eager execution."""
import copy
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def setup(name="tiny_light_c1.pt", **over):
    from causal_gen_amd import vae
    from causal_gen_amd.hps import Hparams

    fx = load_golden(name)
    hpd = dict(fx["hp"])
    hpd.update(lr=2e-3, lr_warmup_steps=2, wd=0.05, beta=2.0)
    hpd.update(over)
    m = vae.HVAE(Hparams(**hpd))
    m.load_state_dict(fx["state_dict"])
    return fx, hpd, m.cuda()


def test_train_steps_match_oracle():
    from causal_gen_amd.train import TrainStep
    from oracle import train_ref

    fx, hpd, m = setup()
    hp = SimpleNamespace(**hpd)
    ref = train_ref.RefTrainer(fx["state_dict"], hp)
    ts = TrainStep(m, hp, ema=True, use_graph=False)
    x, pa = fx["x"], fx["pa"]
    g = torch.Generator().manual_seed(0)
    for step in range(5):
        eps = [torch.randn(e.shape, generator=g) for e in fx["fwd"]["eps"]]
        r_out, r_gn = ref.step(x, pa, noise=[e.clone() for e in eps])
        m.noise = [e.clone() for e in eps]
        out = ts.step(x.cuda(), pa.cuda())
        st = ts.stats()
        got = [float(v) for v in out.cpu()]
        for a, b in zip(got, (r_out["elbo"], r_out["nll"], r_out["kl"])):
            assert abs(a - b) / abs(b) < 2e-4, (step, got, r_out)
        assert abs(st["grad_norm"] - r_gn) / r_gn < 2e-3, (step, st, r_gn)
        assert st["opt_steps"] == ref.opt_steps
    sd = m.state_dict()
    ema = ts.ema_model.state_dict()
    for k, v in ref.sd.items():
        torch.testing.assert_close(sd[k].cpu(), v.detach(), rtol=2e-3, atol=2e-5, msg=lambda s: f"{k}: {s}")
        torch.testing.assert_close(ema[k].cpu(), ref.ema[k], rtol=2e-3, atol=2e-5, msg=lambda s: f"ema {k}: {s}")


def test_skip_on_huge_gradient_leaves_everything_untouched():
    from causal_gen_amd.train import TrainStep

    fx, hpd, m = setup(grad_skip=1e-6)  # every step is "too large"
    ts = TrainStep(m, SimpleNamespace(**hpd), ema=True, use_graph=False)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    for _ in range(2):
        ts.step(fx["x"].cuda(), fx["pa"].cuda())
    st = ts.stats()
    assert st["n_skipped"] == 2 and st["opt_steps"] == 0 and st["skipped_last"]
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])


def test_graph_replay_equals_eager():
    from causal_gen_amd.train import TrainStep

    outs = []
    for use_graph in (False, True):
        fx, hpd, m = setup()
        torch.manual_seed(123)
        ts = TrainStep(m, SimpleNamespace(**hpd), ema=True, use_graph=use_graph)
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        for _ in range(4):
            o = ts.step(x, pa)
        torch.cuda.synchronize()
        outs.append(([float(v) for v in o.cpu()], {k: v.clone() for k, v in m.state_dict().items()}, ts.stats()))
    (o0, s0, t0), (o1, s1, t1) = outs
    assert t0["opt_steps"] == t1["opt_steps"] == 4
    assert o0 == o1, (o0, o1)  # same kernels, same addresses, same Philox counters => bitwise equal
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.parametrize("name", ["simple_vae_c1.pt", "simple_vae_gauss1.pt", "simple_vae_dmol3.pt"])
def test_fused_train_step_on_the_config1_model(name):
    """TrainStep drives simple_vae.VAE like the HVAE (same engine): the first step reports the autograd path's ELBO, the
    parameters move, and hipGraph replay equals eager execution bit for bit -- including GaussNet's device-side
    dequantisation noise, whose Philox snapshot the captured backward pass reads again."""
    from causal_gen_amd import simple_vae
    from causal_gen_amd.hps import Hparams
    from causal_gen_amd.train import TrainStep

    fx = load_golden(name)
    hpd = {k: v for k, v in fx["hp"].items() if k != "hidden_dim"}
    hpd.update(lr=1e-3, betas=(0.9, 0.9), wd=0.01, grad_clip=350.0, grad_skip=5000.0, ema_rate=0.99, lr_warmup_steps=2, beta=1.0,
               accu_steps=1, kl_free_bits=0.0)
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    outs = []
    for use_graph in (False, True):
        m = simple_vae.VAE(Hparams(**hpd))
        m.load_state_dict(fx["state_dict"])
        m = m.cuda().train()
        torch.manual_seed(123)
        ts = TrainStep(m, SimpleNamespace(**hpd), ema=True, use_graph=use_graph)
        first = None
        for _ in range(4):
            o = ts.step(x, pa)
            first = o.clone() if first is None else first
        torch.cuda.synchronize()
        outs.append(([float(v) for v in o.cpu()], [float(v) for v in first.cpu()], {k: v.clone() for k, v in m.state_dict().items()}, ts.stats()))
    (o0, f0, s0, t0), (o1, f1, s1, t1) = outs
    assert t0["opt_steps"] == t1["opt_steps"] == 4 and t0["n_skipped"] == 0
    assert o0 == o1 and f0 == f1, (o0, o1)
    assert all(torch.isfinite(torch.tensor(o0))) and o0[0] < f0[0]  # four steps on one batch lower its ELBO
    moved = 0
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
        moved += int(not torch.equal(s0[k].cpu(), fx["state_dict"][k]))
    assert moved > 10


def test_counterfactual_api_and_dscm_forward():
    from causal_gen_amd import dscm

    fx, hpd, m = setup("tiny_default_c1.pt")
    m.eval()
    x, pa, cf = fx["x"].cuda(), fx["pa"].cuda(), fx["cf_pa"].cuda()
    ab = fx["abduct"]
    m.noise = [e.clone() for e in ab["eps"]]
    cf_x = dscm.counterfactual(m, x, pa, cf, t_abduct=ab["t"])
    ok = fx["cf"]["rec_scale"] > 1e-3
    assert ((cf_x.cpu() - fx["cf"]["cf_x"]).abs()[ok]).max() < 1e-3
    # null intervention => the counterfactual is the observation (up to the clamp), whatever the noise
    m.noise = None
    same = dscm.counterfactual(m, x, pa, pa)
    assert (same - x).abs().max() < 1e-4

    class StubPGM(torch.nn.Module):
        def counterfactual(self, obs, intervention, num_particles=1):
            return {k: intervention.get(k, v) for k, v in obs.items()}

    args = SimpleNamespace(**hpd, parents_x=["a", "b", "c"], dataset="none", lmbda_init=1.0, elbo_constraint=2.0, damping=10.0)
    model = dscm.DSCM(args, StubPGM(), None, m)
    obs = {"x": x, "a": pa[:, 0, 0, 0], "b": pa[:, 1, 0, 0], "c": pa[:, 2:3, 0, 0]}
    do = {"a": cf[:, 0, 0, 0]}
    out = model(obs, do, None, cf_particles=3, t_abduct=0.9)
    assert out["cfs"]["x"].shape == x.shape and out["var_cf_x"].shape == x.shape
    assert torch.isfinite(out["cfs"]["x"]).all() and (out["var_cf_x"] >= -1e-6).all()
    assert dscm.vae_preprocess(args, {"a": torch.ones(2, 1), "b": torch.ones(2, 1), "c": torch.ones(2, 1)}).shape == (2, 3, 16, 16)


def test_backward_accumulates_into_existing_grads():
    """torch semantics on the drop-in surface: a second backward without zero_grad() adds to .grad (what
    trainer.py:64-67 relies on for accu_steps > 1)."""
    fx, hpd, m = setup()
    m.eval()
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    g = torch.Generator().manual_seed(3)
    eps = [[torch.randn(e.shape, generator=g) for e in fx["fwd"]["eps"]] for _ in range(2)]
    singles = []
    for k in range(2):
        m.zero_grad(set_to_none=True)
        m.noise = [e.clone() for e in eps[k]]
        (m(x, pa, beta=1.5)["elbo"] / 2).backward()
        singles.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    m.zero_grad(set_to_none=True)
    for k in range(2):
        m.noise = [e.clone() for e in eps[k]]
        (m(x, pa, beta=1.5)["elbo"] / 2).backward()
    for n, p in m.named_parameters():
        if n in singles[0]:
            want = singles[0][n] + singles[1][n]
            torch.testing.assert_close(p.grad, want, rtol=1e-5, atol=1e-7, msg=lambda s: f"{n}: {s}")


def test_gradient_accumulation_matches_oracle():
    """accu_steps = 2 (trainer.py:62-87): optimiser on iterations 0, 2, 4 over the summed elbo/2 gradients."""
    from causal_gen_amd.train import TrainStep
    from oracle import train_ref

    fx, hpd, m = setup(accu_steps=2)
    hp = SimpleNamespace(**hpd)
    ref = train_ref.RefTrainer(fx["state_dict"], hp)
    ts = TrainStep(m, hp, ema=True, use_graph=False)
    x, pa = fx["x"], fx["pa"]
    g = torch.Generator().manual_seed(0)
    for i in range(5):
        eps = [torch.randn(e.shape, generator=g) for e in fx["fwd"]["eps"]]
        r_out, r_gn = ref.iteration(i, x, pa, noise=[e.clone() for e in eps])
        m.noise = [e.clone() for e in eps]
        out = ts.step(x.cuda(), pa.cuda())
        got = [float(v) for v in out.cpu()]
        for a, b in zip(got, (r_out["elbo"], r_out["nll"], r_out["kl"])):
            assert abs(a - b) / abs(b) < 2e-4, (i, got, r_out)
        st = ts.stats()
        assert st["opt_steps"] == ref.opt_steps, (i, st, ref.opt_steps)
        if r_gn is not None:
            assert abs(st["grad_norm"] - r_gn) / r_gn < 2e-3, (i, st, r_gn)
    assert ref.opt_steps == 3
    sd = m.state_dict()
    for k, v in ref.sd.items():
        torch.testing.assert_close(sd[k].cpu(), v.detach(), rtol=2e-3, atol=2e-5, msg=lambda s: f"{k}: {s}")


def test_gradient_accumulation_graph_equals_eager():
    """The accumulate-only and the stepping iteration are two hipGraphs; replaying them must equal eager execution."""
    from causal_gen_amd.train import TrainStep

    outs = []
    for use_graph in (False, True):
        fx, hpd, m = setup(accu_steps=2)
        torch.manual_seed(321)
        ts = TrainStep(m, SimpleNamespace(**hpd), ema=True, use_graph=use_graph)
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        for _ in range(7):
            o = ts.step(x, pa)
        torch.cuda.synchronize()
        outs.append(([float(v) for v in o.cpu()], {k: v.clone() for k, v in m.state_dict().items()}, ts.stats()))
    (o0, s0, t0), (o1, s1, t1) = outs
    assert t0["opt_steps"] == t1["opt_steps"] == 4  # iterations 0, 2, 4, 6
    assert o0 == o1, (o0, o1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def _free_port():
    import socket

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return str(s_.getsockname()[1])


def test_dp_two_ranks_reproduce_the_single_process_step():
    """Two ranks x B/2 through the default graph-captured data-parallel TrainStep == one process x B after three optimiser steps
    (tests/dp_trainstep_worker.py): the rank-averaged flat gradient is the gradient of the global-batch mean on the real path."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "tests", "dp_trainstep_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DP_TRAINSTEP_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("launcher", ["torch.distributed.run", "self"])
def test_dp_bench_path_two_gloo_ranks_on_one_gpu(launcher):
    """The data-parallel step (hipGraph fwd+bwd -> bucketed gradient all-reduce -> hipGraph step tail) end to end: two
    ranks share this GPU over gloo (RCCL wants one device per rank; the code path is the same).  Both ranks must finish
    and rank 0 must print one JSON line with n_gpus = 2 and a finite ELBO.  "self": the plain `python bench.py --gpus 2`
    command starts its own ranks."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CGEN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    args = ["--gpus", "2", "--config", "morphomnist", "--batch", "16", "--steps", "2", "--warmup", "1", "--prep-steps", "1", "--no-cf"]
    if launcher == "self":
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", _free_port(), os.path.join(root, "bench.py")] + args
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["elbo_nats_per_dim"] == d["elbo_nats_per_dim"]
    assert d["opt_steps"] == 4 and "roofline" in d
    assert d["dp"]["gloo_ranks_seen"] == 2 and d["dp"]["gradient_bytes"] > 0


def test_dp_allreduce_overlapped_with_backward_equals_the_serialized_exchange():
    """north_star: gradient all-reduce overlapped with backward.  The backward graph is cut where the decoder half of the
    flat gradient is final; that half travels on the communication stream under the encoder half.  Two gloo ranks on this
    GPU: parameters after four steps are bit-equal to the serialized exchange (a two-rank sum is order independent), the
    overlapped path really ran (early bytes > 0), and the exposed communication time is reported."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for overlap, port in (("0", _free_port()), ("1", _free_port())):
        env = dict(os.environ, CGEN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", CGEN_DP_OVERLAP=overlap)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "ukbb192", "--batch", "2",
               "--steps", "2", "--warmup", "1", "--prep-steps", "1", "--no-cf", "--no-cpu"]
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[overlap] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    a, b = outs["0"], outs["1"]
    assert not a["dp"]["allreduce_overlapped_with_backward"] and b["dp"]["allreduce_overlapped_with_backward"]
    assert b["dp"]["early_bytes"] > 0 and b["dp"]["late_bytes"] > 0 and b["dp"]["exposed_comm_ms_per_step"] is not None
    assert a["param_abs_sum"] == b["param_abs_sum"], (a["param_abs_sum"], b["param_abs_sum"])
    assert a["elbo_nats_per_dim"] == b["elbo_nats_per_dim"] and a["opt_steps"] == b["opt_steps"] == 4


def test_backward_is_bit_reproducible_next_to_the_background_weight_gradient_kernel():
    """Three backward passes of the ukbb192 model (bf16, background flush on) from identical state must give bit-identical
    gradients.  They did not while packed-f32 instructions were in the kernels: a co-resident MFMA wave of the background
    weight-gradient kernel corrupts them (tools/coexec_probe.py); build.sh keeps them out of every kernel."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    m, hp = bench.build_model("ukbb192", "f16")
    m = m.cuda().train()
    x, pa = bench.synth_batch("ukbb192", hp, 8, "cuda", 1)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    eng = m.engine()
    eng.rng_ptr()
    assert eng.wgrad_flush_frac, "the background flush is the default"
    runs = []
    for _ in range(3):
        m.zero_grad()
        eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
        out = m(x, pa, beta=1.0)
        out["elbo"].backward()
        torch.cuda.synchronize()
        runs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    for r in runs[1:]:
        bad = [n for n in runs[0] if not torch.equal(runs[0][n], r[n])]
        assert not bad, (len(bad), bad[:4])
    # the launch-structure features change WHERE and WHEN work is launched, never the arithmetic: switching the background
    # flush, the reparam rider and the side-stream reduce off gives the same bits
    eng.wgrad_flush_frac, eng.ride, eng.wgrad_bg_reduce = [], False, False
    m.zero_grad()
    eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
    out = m(x, pa, beta=1.0)
    out["elbo"].backward()
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.equal(runs[0][n], p.grad)]
    assert not bad, (len(bad), bad[:4])
    # ... and the pair launch of a decoder layer's two fused data gradients (cgen_block3_pair, on by default): it was on in every run
    # above; off gives the same bits
    assert eng.blk3_pair and eng.blk3_pairs > 0, "the posterior / prior data gradients of the 24^2 and 48^2 layers share launches"
    assert eng.conv_pair and eng.conv_pairs > 0, "... and so do small-image data-gradient convs (cgen_conv2d_pair)"
    eng.blk3_pair = eng.conv_pair = False
    m.zero_grad()
    eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
    out = m(x, pa, beta=1.0)
    out["elbo"].backward()
    torch.cuda.synchronize()
    assert eng.blk3_pairs == 0 and eng.conv_pairs == 0
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.equal(runs[0][n], p.grad)]
    assert not bad, ("paired data-gradient launches differ from single ones", len(bad), bad[:4])
    eng.blk3_pair = eng.conv_pair = True


def test_full_size_bf16_path_agrees_with_the_f32_parity_path():
    """ukbb192 at full size (the shapes the lean bf16 kernels are selected for; the golden fixtures are tiny models):
    the throughput path and the f32 parity path must agree on the ELBO and on every parameter gradient's direction."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    res = {}
    for dt in ("f32", "f16"):
        m, hp = bench.build_model("ukbb192", dt)
        m = m.cuda().train()
        g = torch.Generator().manual_seed(3)
        with torch.no_grad():
            for p in m.parameters():
                p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
        x, pa = bench.synth_batch("ukbb192", hp, 2, "cuda", 1)
        eng = m.engine()
        eng.rng_ptr()
        eng.rng.copy_(torch.tensor([11, 0], dtype=torch.int64, device=eng.rng.device))
        out = m(x, pa, beta=1.0)
        out["elbo"].backward()
        torch.cuda.synchronize()
        res[dt] = ({k: float(out[k]) for k in ("elbo", "nll", "kl")},
                   {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None})
        del m, eng
    for k in ("elbo", "nll", "kl"):
        a, b = res["f32"][0][k], res["f16"][0][k]
        assert abs(a - b) <= 3e-3 * abs(a), (k, a, b)
    errs, cosines = [], []
    for n, gf in res["f32"][1].items():
        gb = res["f16"][1][n]
        den = float(gf.norm())
        if den == 0:
            continue
        errs.append(float((gb - gf).norm()) / den)
        cosines.append(float((gb * gf).sum()) / (den * float(gb.norm()) + 1e-30))
    errs.sort()
    assert errs[len(errs) // 2] < 0.02 and errs[-1] < 0.2 and min(cosines) > 0.985, (errs[len(errs) // 2], errs[-1], min(cosines))


def test_full_size_graph_replay_equals_eager_bf16():
    """ukbb192, bf16, four optimiser steps: the captured step (background flush on a side stream, reparam riders, the
    two-stream forward, step tail) replays bit for bit what the eager step computes."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from causal_gen_amd.train import TrainStep

    outs = []
    for use_graph in (False, True):
        m, hp = bench.build_model("ukbb192", "f16")
        m = m.cuda()
        torch.manual_seed(123)
        ts = TrainStep(m, hp, ema=True, use_graph=use_graph)
        x, pa = bench.synth_batch("ukbb192", hp, 4, "cuda", 1)
        for _ in range(4):
            o = ts.step(x, pa)
        torch.cuda.synchronize()
        outs.append(([float(v) for v in o.cpu()], {k: v.detach().clone() for k, v in m.state_dict().items()}, ts.stats()))
        del m, ts
    (o0, s0, t0), (o1, s1, t1) = outs
    assert t0["opt_steps"] == t1["opt_steps"] == 4 and o0 == o1, (o0, o1)
    bad = [k for k in s0 if not torch.equal(s0[k], s1[k])]
    assert not bad, (len(bad), bad[:4])


def test_full_size_null_intervention_returns_the_observation():
    """Size-independent property of abduct -> replay x2 -> dscm.py:55-56: with cf_parents == parents the counterfactual is the
    observation itself, here on ukbb192 in bf16; a real intervention changes the image."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from causal_gen_amd import dscm

    m, hp = bench.build_model("ukbb192", "f16")
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn(p.shape, generator=g).cuda() * 0.02)
    x, pa = bench.synth_batch("ukbb192", hp, 2, "cuda", 1)
    with torch.no_grad():
        same = dscm.counterfactual(m, x, pa, pa)
        diff = dscm.counterfactual(m, x, pa, pa.roll(1, 0))
    assert float((same - x).abs().max()) < 1e-5
    assert float((diff - x).abs().mean()) > 1e-3


def test_free_bits_under_data_parallelism():
    """kl_free_bits > 0 with two ranks: the per-channel KL sums are all-reduced inside the forward pass, so the
    floored KL and the rank-averaged gradients equal the single-process result on the concatenated batch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "tests", "dp_free_bits_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    assert "DP_FREE_BITS_OK" in r.stdout, r.stdout[-1500:]


def test_graphed_counterfactual_matches_eager_and_follows_weight_updates():
    from causal_gen_amd import dscm

    fx, hpd, m = setup("tiny_condprior_morpho_c1.pt")
    m.eval()
    x, pa, cf = fx["x"].cuda(), fx["pa"].cuda(), fx["cf_pa"].cuda()
    gcf = dscm.GraphedCounterfactual(m)

    def reset():
        eng = m.engine()
        eng.rng_ptr()
        eng.rng.copy_(torch.tensor([123, 0], dtype=torch.int64, device=eng.rng.device))  # in place: the graph holds its address

    def both():
        reset()
        a = dscm.counterfactual(m, x, pa, cf).clone()
        reset()
        b = gcf(x, pa, cf).clone()
        return a, b

    a0, b0 = both()      # first graphed call = eager warm-up + capture
    a1, b1 = both()      # replay
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.05)
    a2, b2 = both()
    assert torch.equal(a2, b2) and not torch.equal(a2, a1)


def test_inference_after_train_steps_sees_the_updated_weights():
    """The fused AdamW / EMA kernel writes the flat parameter buffers through raw pointers (no torch version bump): an
    inference call after TrainStep.step() -- validation on ema_model (trainer.py:39-40,94), sampling, counterfactuals --
    must re-image the weights.  ema_model(x) after N steps == a fresh model loaded from ema_model.state_dict()."""
    from causal_gen_amd import vae
    from causal_gen_amd.hps import Hparams
    from causal_gen_amd.train import TrainStep

    for use_graph in (False, True):
        fx, hpd, m = setup()
        hp = SimpleNamespace(**hpd)
        ts = TrainStep(m, hp, ema=True, use_graph=use_graph)
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        eps = [e.clone() for e in fx["fwd"]["eps"]]

        def evaluate(model):
            model.noise = [e.clone() for e in eps]
            with torch.no_grad():
                o = model(x, pa, beta=1.0)
            model.noise = None
            return [float(o[k]) for k in ("elbo", "nll", "kl")]

        seen = []
        for rounds in range(2):  # train -> eval -> train -> eval
            for _ in range(3):
                ts.step(x, pa)
            for live in (ts.ema_model, m):
                was = live.training
                live.eval()
                got = evaluate(live)
                fresh = vae.HVAE(Hparams(**hpd))
                fresh.load_state_dict({k: v.clone() for k, v in live.state_dict().items()})
                want = evaluate(fresh.cuda().eval())
                live.train(was)
                assert got == want, (use_graph, rounds, got, want)
                seen.append(got[0])
        assert seen[0] != seen[2] and seen[1] != seen[3], seen  # the numbers move from round to round (EMA == model during its warm-up)


def test_beta_warmup_replays_one_graph_and_matches_eager():
    """beta warm-up (trainer.py:57): beta is device data of the captured step, so the schedule replays ONE graph instead of
    capturing (and leaking) a graph per distinct beta; results equal the eager path's bit for bit."""
    from causal_gen_amd.train import TrainStep

    res = []
    for use_graph in (False, True):
        fx, hpd, m = setup(beta_warmup_steps=4)
        hp = SimpleNamespace(**hpd)
        torch.manual_seed(5)
        ts = TrainStep(m, hp, ema=True, use_graph=use_graph)
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        outs = []
        for _ in range(7):
            outs.append([float(v) for v in ts.step(x, pa).cpu()])
        if use_graph:
            assert len(ts.graphs) == 1, list(ts.graphs)
        res.append((outs, {k: v.clone() for k, v in m.state_dict().items()}))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    assert all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])
    elbos = [o[0] - o[1] for o in res[0][0]]  # beta * kl grows with the schedule while kl itself stays O(1)
    assert elbos[0] < elbos[3]


def test_two_batch_shapes_keep_their_own_step_constants():
    """Alternating batch shapes replays graphs that each hold the address of THEIR gradient-seed constants."""
    from causal_gen_amd.train import TrainStep

    res = []
    for use_graph in (False, True):
        fx, hpd, m = setup()
        hp = SimpleNamespace(**hpd)
        torch.manual_seed(5)
        ts = TrainStep(m, hp, ema=False, use_graph=use_graph)
        x, pa = fx["x"].cuda(), fx["pa"].cuda()
        outs = []
        for i in range(6):
            n = 3 if i % 2 == 0 else 2
            outs.append([float(v) for v in ts.step(x[:n].contiguous(), pa[:n].contiguous()).cpu()])
        res.append((outs, {k: v.clone() for k, v in m.state_dict().items()}))
    assert res[0][0] == res[1][0]
    assert all(torch.equal(res[0][1][k], res[1][1][k]) for k in res[0][1])


def test_train_step_state_dict_resumes_moments_and_schedules(tmp_path):
    """main.py:75-90: a resumed run continues Adam's moments, the LR warm-up and the EMA warm-up.  5 steps == 3 steps ->
    checkpoint (reference wire format, AdamW-layout optimiser state) -> fresh process state -> 2 steps."""
    from causal_gen_amd import checkpoint, vae
    from causal_gen_amd.hps import Hparams
    from causal_gen_amd.train import TrainStep

    fx, hpd, m = setup(lr_warmup_steps=4)
    hp = SimpleNamespace(**hpd)
    x, pa = fx["x"].cuda(), fx["pa"].cuda()
    g = torch.Generator().manual_seed(0)
    eps = [[torch.randn(e.shape, generator=g) for e in fx["fwd"]["eps"]] for _ in range(5)]
    ts = TrainStep(m, hp, ema=True, use_graph=False)
    for s in range(3):
        m.noise = [e.clone() for e in eps[s]]
        ts.step(x, pa)
    path = str(tmp_path / "checkpoint.pt")
    checkpoint.save_checkpoint(path, m, ts.ema_model, Hparams(**hpd), epoch=1, step=3, train_step=ts)
    ck = torch.load(path, weights_only=False)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p)) for p in m.parameters()], lr=hp.lr, betas=tuple(hp.betas),
                            weight_decay=hp.wd)
    opt.load_state_dict({k: v for k, v in ck["optimizer_state_dict"].items() if k != "cgen"})  # a stock AdamW accepts the layout
    for s in range(3, 5):
        m.noise = [e.clone() for e in eps[s]]
        ts.step(x, pa)
    want = ({k: v.clone() for k, v in m.state_dict().items()}, {k: v.clone() for k, v in ts.ema_model.state_dict().items()})
    m2 = vae.HVAE(Hparams(**hpd)).cuda()
    ts2 = TrainStep(m2, hp, ema=True, use_graph=False)
    info = checkpoint.resume_train_step(path, ts2)
    assert info["step"] == 3 and ts2.stats()["opt_steps"] == 3
    for s in range(3, 5):
        m2.noise = [e.clone() for e in eps[s]]
        ts2.step(x, pa)
    for k, v in want[0].items():
        assert torch.equal(m2.state_dict()[k], v), k
    for k, v in want[1].items():
        assert torch.equal(ts2.ema_model.state_dict()[k], v), k


def test_shared_covariance_bias_gets_its_gradient():
    """x_like='shared_dgauss' with std_init > 0 (vae.py:335-345): likelihood.x_logscale.weight is frozen, its bias trains."""
    from causal_gen_amd import vae
    from causal_gen_amd.hps import Hparams
    from oracle import hvae_ref

    fx = load_golden("tiny_light_c1.pt")
    hpd = dict(fx["hp"])
    hpd.update(x_like="shared_dgauss", std_init=0.3)
    m = vae.HVAE(Hparams(**hpd))
    sd0 = {k: v.clone() for k, v in fx["state_dict"].items()}
    sd0["likelihood.x_logscale.weight"] = torch.zeros_like(sd0["likelihood.x_logscale.weight"])
    sd0["likelihood.x_logscale.bias"] = torch.full_like(sd0["likelihood.x_logscale.bias"], float(torch.tensor(0.3).log()))
    m.load_state_dict(sd0)
    m = m.cuda().eval()
    assert not m.likelihood.x_logscale.weight.requires_grad and m.likelihood.x_logscale.bias.requires_grad
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    eps = fx["fwd"]["eps"]
    ref = hvae_ref.hvae_forward(sd, SimpleNamespace(**hpd), fx["x"], fx["pa"], beta=1.0, noise=hvae_ref._Noise([e.clone() for e in eps]))
    ref["elbo"].backward()
    m.noise = [e.clone() for e in eps]
    out = m(fx["x"].cuda(), fx["pa"].cuda(), beta=1.0)
    out["elbo"].backward()
    torch.cuda.synchronize()
    gb, rb = m.likelihood.x_logscale.bias.grad, sd["likelihood.x_logscale.bias"].grad
    assert gb is not None and float((gb.cpu() - rb).abs().max()) < 2e-3 * float(rb.abs().max())
    assert m.likelihood.x_logscale.weight.grad is None


def test_virtual_parents_equal_materialised():
    """parents handed over as the stride-0 expand() view (or [B,ctx]) are laid out as [B,1,1,ctx] and
    broadcast by stride; ELBO, gradients and decoded means must be BIT-identical to the materialised [B,ctx,R,R] tensor
    (same kernels, same K order, same values), including the drop_cond scaling of a conditional prior."""
    from causal_gen_amd import vae
    from causal_gen_amd.hps import setup_hparams

    for name, B, dt in (("ukbb192", 2, "f32"), ("ukbb192", 2, "f16"), ("morphomnist", 4, "f32"), ("morphomnist", 4, "f16")):
        hp = setup_hparams(name)
        torch.manual_seed(3)
        m = vae.HVAE(hp).cuda()
        m.compute_dtype = dt
        m.train()
        if m.cond_prior:
            m.decoder.__dict__["drop_cond"] = lambda: (0, 1)  # the draw that rescales the parents (vae.py:244-247)
        g = torch.Generator().manual_seed(5)
        x = ((torch.randint(0, 256, (B, hp.input_channels, hp.input_res, hp.input_res), generator=g).float() - 127.5) / 127.5).cuda()
        pc = torch.randn(B, hp.context_dim, generator=g).cuda()
        R = hp.input_res
        forms = {"repeat": pc[..., None, None].repeat(1, 1, R, R), "expand": pc[..., None, None].expand(-1, -1, R, R), "flat": pc}

        def eps():
            ge = torch.Generator().manual_seed(100)
            return [torch.randn(B, b.z_dim, b.res, b.res, generator=ge) for b in m.decoder.blocks if b.stochastic]

        res = {}
        for k, pa in forms.items():
            m.zero_grad(set_to_none=True)
            m.noise = eps()
            out = m(x, pa, beta=hp.beta)
            assert not m.noise
            out["elbo"].backward()
            grads = torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
            m.noise = eps()
            with torch.no_grad():
                zs = m.abduct(x, pa)
                loc, _ = m.forward_latents(zs, pa)
            m.noise = None
            res[k] = (out["elbo"].detach().clone(), grads.clone(), loc.clone())
        assert float(res["repeat"][1].abs().sum()) > 0
        for k in ("expand", "flat"):
            for i, what in enumerate(("elbo", "grads", "decoded mean")):
                assert torch.equal(res[k][i], res["repeat"][i]), (name, dt, k, what)


def test_f16_training_tracks_the_f32_path():
    """Loss-scaled binary16 training against the f32 parity path: two models from one init, the same stream of fresh
    synthetic batches and the same Philox noise through the captured `TrainStep` (AdamW, warm-up, clip / skip, EMA).  After 40
    optimiser steps the ELBO curves agree to 1e-4 at every step, no step was skipped on either path, the parameters are within
    2 % (relative L2).  tools/train_track.py is the long form (300 steps: 4e-5 on ukbb192, 3e-6 on morphomnist)."""
    import bench
    from causal_gen_amd.train import TrainStep

    curves, finals, skipped = {}, {}, {}
    for dt in ("f32", "f16"):
        m, hp = bench.build_model("morphomnist", dt)
        m = m.cuda().train()
        m.decoder.__dict__["drop_cond"] = lambda: (1, 1)
        ts = TrainStep(m, hp, ema=True, use_graph=True)
        eng = m.engine()
        eng.rng_ptr()
        eng.rng.copy_(torch.tensor([77, 0], dtype=torch.int64, device=eng.rng.device))
        c = []
        for it in range(40):
            x, pa = bench.synth_batch("morphomnist", hp, 16, "cuda", 500 + it)
            c.append(ts.step(x, pa).clone())
        torch.cuda.synchronize()
        curves[dt] = torch.stack(c).cpu()
        finals[dt] = torch.cat([p.detach().flatten().float() for p in m.parameters()]).cpu()
        skipped[dt] = ts.stats()["n_skipped"]
        if dt == "f16":
            assert eng.loss_scale >= 1024.0  # (16 * 1024 terms -> 2^13)
    assert skipped == {"f32": 0, "f16": 0}, skipped
    rel = ((curves["f16"][:, 0] - curves["f32"][:, 0]).abs() / curves["f32"][:, 0].abs()).max()
    assert float(rel) < 1e-4, float(rel)
    assert float(curves["f32"][-1, 0]) < float(curves["f32"][0, 0])  # (it did train)
    d = float((finals["f16"] - finals["f32"]).norm() / finals["f32"].norm())
    assert d < 2e-2, d


def test_f16_loss_scale_backs_off_inside_step_and_grows_back():
    """the binary16 loss-scale back-off lives in TrainStep.step() (all ranks, keyed by the iteration counter), not in
    the reporting getter.  An absurd initial scale (2^30) overflows every activation gradient: every step is dropped for a
    non-finite norm until enough checks have halved it; each check halves ONCE, rebuilds the captured graph, stats() changes
    nothing, the shift survives a state_dict round trip, and clean checks grow the scale back (never above the rule)."""
    import bench
    from causal_gen_amd.train import TrainStep

    m, hp = bench.build_model("morphomnist", "f16")
    m = m.cuda().train()
    m.decoder.__dict__["drop_cond"] = lambda: (1, 1)
    ts = TrainStep(m, hp, ema=False, use_graph=True)
    ts.ls_check_interval, ts.ls_growth_interval = 2, 3
    eng = m.engine()
    rule = eng.loss_scale_rule(16 * 1024)
    x, pa = bench.synth_batch("morphomnist", hp, 16, "cuda", 900)
    ts.step(x, pa)
    assert eng.loss_scale == rule and eng.loss_scale_shift == 0
    # poison: a shift ABOVE the rule cannot be set through _rescale (clamped to <= 0) -- emulate a spike by raising the rule itself
    os.environ["CGEN_LOSS_SCALE_LOG2"] = "30"
    try:
        ts._rescale(0)  # drop graphs / coefficient tables: the next step bakes 2^30 in
        shifts, graphs_seen = [], set()
        for it in range(120):
            ts.step(x, pa)
            st0 = (eng.loss_scale_shift, ts.overflow_backoffs)
            s1, s2 = ts.stats(), ts.stats()  # read-only, however often it is called
            assert (eng.loss_scale_shift, ts.overflow_backoffs) == st0 and repr(s1) == repr(s2)
            shifts.append(eng.loss_scale_shift)
            graphs_seen.add(id(next(iter(ts.graphs.values()))[0]) if ts.graphs else None)
            if not s1["skipped_last"] and eng.loss_scale_shift < 0:
                break
        assert shifts[-1] < 0 and not ts.stats()["skipped_last"], (shifts, ts.stats())
        d = [b - a for a, b in zip(shifts, shifts[1:])]
        assert all(v in (0, -1) for v in d), d                      # one halving per check at most
        assert not any(a == -1 and b == -1 for a, b in zip(d, d[1:]))  # ... and only on check iterations (interval 2)
        assert ts.overflow_backoffs == -shifts[-1] >= 1
        assert len(graphs_seen - {None}) >= 2                     # the captured step was rebuilt after a halving
        assert ts.stats()["n_skipped"] >= 2
        sd = ts.state_dict()
        assert sd["cgen"]["loss_scale_shift"] == shifts[-1]
    finally:
        del os.environ["CGEN_LOSS_SCALE_LOG2"]
    # the spike is over (the rule is back): a restored TrainStep carries the shift, clean checks then grow it back to 0, not beyond
    m2, _ = bench.build_model("morphomnist", "f16")
    m2 = m2.cuda().train()
    m2.decoder.__dict__["drop_cond"] = lambda: (1, 1)
    m2.load_state_dict(m.state_dict())
    ts2 = TrainStep(m2, hp, ema=False, use_graph=True)
    ts2.ls_check_interval, ts2.ls_growth_interval = 2, 3
    ts2.load_state_dict(sd)
    e2 = m2.engine()
    assert e2.loss_scale_shift == shifts[-1]
    seen = []
    for it in range(2 * 3 * (-shifts[-1]) + 8):
        ts2.step(x, pa)
        seen.append(e2.loss_scale_shift)
    assert seen[-1] == 0 and max(seen) == 0 and all(b - a in (0, 1) for a, b in zip(seen, seen[1:])), seen
    assert e2.loss_scale == rule and ts2.stats()["skipped_last"] is False
