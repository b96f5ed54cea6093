"""The per-step training harness of the reference (trainer.py:49-87, train_setup.py:42-53, utils.py:87-228) as a
fixed launch sequence on the GPU:

    weight images -> HVAE forward -> hand-written backward -> [DP: gradient all-reduce] -> global grad-norm ->
    clip / NaN-or-norm skip predicate (on device) -> fused AdamW + EMA -> step counter commit

No autograd, no host synchronisation inside the step (the reference has three): the skip decision,
the LR warm-up, Adam's bias correction and the EMA warm-up schedule are all evaluated on the device from a
device-side count of successful steps.  The whole sequence is captured in a hipGraph after one eager step.
Under data parallelism (one process per GPU, RCCL over xGMI) the flat f32 gradient is averaged across ranks in
buckets between the backward and the norm; ranks share the seed-driven draws and the skip flag by construction
(the flag is a function of the all-reduced gradient and of the all-reduced nll/kl).
"""
import copy
import math
import ctypes as C

import torch

from . import _lib, dp
from .engine import StaticParents, compact_parents


def preprocess_batch(args, batch, expand_pa=False):
    """trainer.py:16-21 on the device: u8 pixels -> [-1, 1] f32 (one HIP launch, no ATen arithmetic), parents to f32 and,
    for the HVAE's parent concatenation, broadcast to [B, ctx, R, R] -- as a stride-0 ``expand`` view, not a ``repeat``: same
    shape and values for every consumer, and the HVAE then lays out only the [B,1,1,ctx] vector (engine.from_parents).
    ``HVAE.forward`` also accepts the raw u8 batch directly and fuses the normalisation into its NCHW -> NHWC load, which
    skips this pass altogether."""
    lib = _lib.require_gpu()
    dev = torch.device(getattr(args, "device", "cuda"))
    x = batch["x"].to(dev)
    if x.dtype == torch.uint8:
        x = x.contiguous()
        n, c, h, w = x.shape
        out = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)  # channels-last memory, NCHW shape below
        lib.nchw_to_nhwc(1, _lib.F32, n, c, h, w, x.data_ptr(), _lib.View(out.data_ptr(), h * w * c, w * c, c, c, 0), 127.5,
                         1.0 / 127.5, torch.cuda.current_stream(dev).cuda_stream)
        x = out.permute(0, 3, 1, 2)
    else:
        x = (x.float() - 127.5) / 127.5
    batch["x"] = x
    batch["pa"] = batch["pa"].to(dev).float()
    if expand_pa:
        batch["pa"] = batch["pa"][..., None, None].expand(-1, -1, *(args.input_res,) * 2)
    return batch


# hipGraph captures are THREAD-LOCAL: with a ProcessGroupNCCL alive, its watchdog thread polls events (hipEventQuery) at any
# moment, which a capture in the default "global" mode treats as an illegal call and dies of ("operation failed due to a previous
# error during capture") -- found by tools/dp_rccl_dryrun.py, invisible to the gloo tests
CAPTURE_MODE = "thread_local"


def _covers(have, want):
    """True when every [lo, hi) of `want` lies inside one range of `have` (both sorted, disjoint)."""
    return all(any(a <= lo and hi <= b for a, b in have) for lo, hi in want)


def _subtract_ranges(whole, holes):
    """[lo, hi) ranges of `whole` not covered by `holes` (both sorted, non-overlapping)."""
    out = []
    for lo, hi in whole:
        cur = lo
        for a, b in holes:
            if b <= cur or a >= hi:
                continue
            if a > cur:
                out.append((cur, a))
            cur = max(cur, b)
        if cur < hi:
            out.append((cur, hi))
    return out


def linear_warmup(warmup_iters):
    """utils.py:32-36."""
    return lambda it: 1.0 if it > warmup_iters else it / warmup_iters


class TrainStep:
    NORM_BLOCKS = 1024

    def __init__(self, model, args, ema=True, use_graph=True, process_group=None, bucket_mb=32):
        self.model, self.args = model, args
        self.lib = _lib.require_gpu()  # (the step tail's own kernels; the model's launches go through eng.lib, the staging proxy)
        self.use_graph = use_graph
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        import os as _os0
        # the data-parallel machinery (communication stream, split backward graph, bucketed all-reduce) is active with more than
        # one rank -- or, for a dry run of exactly that code through RCCL on ONE GPU, with CGEN_DP_FORCE=1 and a 1-rank group
        self.dp = self.world > 1 or (process_group is not None and _os0.environ.get("CGEN_DP_FORCE") == "1")
        if self.dp and getattr(model, "free_bits", 0) > 0:
            self.use_graph = False  # free bits exchange per-channel KL sums inside the forward pass (vae.py): not capturable
        model.train()
        eng = model.engine()
        self.eng = eng
        n = eng.flat_p.numel()
        dev = eng.device
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.ema_model = None
        if ema:
            self.ema_model = copy.deepcopy(model).to(dev)
            self.ema_model.requires_grad_(False)
            self.ema_model.eval()
            self.ema_flat = self.ema_model.engine().flat_p
        self.state = torch.zeros(8, device=dev)
        self.partial = torch.zeros(self.NORM_BLOCKS, device=dev)
        self.coefs = {}   # (B, dims) -> [device tensor (1/(B dims accu), beta/(B dims accu), beta), beta last written]
        self.coef = None
        self.ranges = None
        self.graphs = {}
        self.static = None
        self.out3 = None
        self.beta = float(args.beta)
        self.it = 0
        # DP: the gradient exchange either follows the backward graph (serialized: graph | bucketed all-reduce | optimiser graph)
        # or travels under it (overlap: the backward graph is cut where the decoder half of the gradient is final, that half is
        # exchanged on the communication stream under the encoder's backward pass).  Overlap is not free: it needs a second
        # background weight-gradient flush (+0.35 ms) and a third graph, +1.0 ms per step on ukbb192 measured on one MI355X
        # through a 1-rank RCCL group (tools/dp_rccl_dryrun.py, bench.py with CGEN_DP_FORCE=1: 18.43 vs 17.39 ms), while the
        # serialized exchange of its 69.5 MB costs ~0.4 ms on an xGMI node (2 (N-1)/N x bytes at ~300 GB/s bus bandwidth).
        # CGEN_DP_OVERLAP=1 / 0 forces a form; unset, the step overlaps only when the estimated exchange exceeds that price.
        import os as _os
        _ov = _os.environ.get("CGEN_DP_OVERLAP")
        if _ov is None:
            est_s = 4.0 * n * 2.0 * max(self.world - 1, 0) / max(self.world, 1) / float(_os.environ.get("CGEN_DP_BUS_GBS", "300")) / 1e9
            self.dp_overlap = self.dp and est_s > 1.0e-3
            self.dp_policy = "auto: estimated exchange %.2f ms %s the ~1 ms the overlapped form costs" % (1e3 * est_s, ">" if self.dp_overlap else "<=")
        else:
            self.dp_overlap = self.dp and _ov != "0"
            self.dp_policy = "forced by CGEN_DP_OVERLAP=" + _ov
        if self.dp_overlap:
            # a second background flush at the end of the DECODER's backward pass (75.5 % of the weight-gradient work of the
            # 192^2 presets; costs 0.8 % on one GPU): from there on every decoder / likelihood gradient is final -- 48 of the
            # 69.5 MB of ukbb192 -- and is exchanged under the encoder's backward pass; the graph is cut at 90 %
            if "CGEN_WGRAD_FLUSH_FRAC" not in _os.environ:
                eng.wgrad_flush_frac = [0.55, 0.755]
            if "CGEN_DP_SPLIT_FRAC" not in _os.environ:
                eng.split_frac = 0.9
        self.comm_stream = None
        self.early_ranges = self.late_ranges = None
        self._early_ok = True
        self.time_comm = False
        self._comm_events = []
        # gradient accumulation (trainer.py:64-67): elbo / accu_steps per iteration, summed in a second flat buffer; the
        # optimiser tail runs on iterations with (it - 1) % accu_steps == 0 and reads that buffer
        self.accu = max(1, int(getattr(args, "accu_steps", 1) or 1))
        # f16 loss-scale back-off: looked after INSIDE step(), on every rank, every `ls_check_interval` iterations,
        # keyed by the iteration counter (idempotent); stats() is read-only.  The decision reads the device state of the
        # previous step, which is identical on all data-parallel ranks (all-reduced gradient + scalars feed clip_decide).
        self.overflow_backoffs, self._clean_checks, self.ls_growth_interval = 0, 0, 20
        self.ls_check_interval, self._ls_checked_it, self._ls_last_growth_it = 16, 0, None
        self._ls_nonfinite_seen = 0.0  # state[6] (steps dropped for a non-finite norm) at the previous check
        self.LS_SHIFT_MIN = -16
        self.acc_g = torch.zeros(n, device=dev) if self.accu > 1 else None

    # -- pieces -----------------------------------------------------------------------------------------
    def _coef_for(self, x, beta):
        """Per-(batch, dims) device constants of a step: gradient seeds d(elbo)/d(sum nll), d(elbo)/d(sum kl) and beta itself.
        They live in device memory so that a captured step follows the beta warm-up (trainer.py:57) without re-capture:
        the host rewrites the three floats (outside any capture) only when beta moved.  One tensor per key, kept alive as
        long as the TrainStep: captured graphs hold its address."""
        B, dims = int(x.shape[0]), float(x[0].numel())
        ent = self.coefs.get((B, dims))
        if ent is None:
            ent = self.coefs[(B, dims)] = [torch.zeros(3, device=self.eng.device), None]
        if ent[1] != beta:
            S = self.eng.set_loss_scale(B * dims * self.accu)  # (1 for f32; divided out again by the parameter-gradient reduces)
            ent[0].copy_(torch.tensor([S / (B * dims * self.accu), S * beta / (B * dims * self.accu), beta], dtype=torch.float32))
            ent[1] = beta
        return ent[0]

    def _fwd_bwd(self, x, pa, beta):
        m, eng = self.model, self.eng
        self.coef = self.coefs[(int(x.shape[0]), float(x[0].numel()))][0]  # written by step() before any capture / replay
        eng.set_loss_scale(int(x.shape[0]) * float(x[0].numel()) * self.accu)
        m.__dict__["_beta_dev"] = self.coef.data_ptr() + 8
        try:
            out3 = m._run_forward(x, pa, beta, record=True)
        finally:
            m.__dict__["_beta_dev"] = None
        params, xin, B, R, Cx, dims = m.__dict__["_saved"]
        eng.kl_coef_ptr = self.coef.data_ptr() + 4
        gparams = eng.seed_grad(params)
        if getattr(m.likelihood, "logit_space", False):
            u, snap = m.__dict__["_gauss_noise"][:2]
            self.lib.gauss_nll_bwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), u, snap.data_ptr(), 977, self.coef.data_ptr(), 0,
                                   gparams.cv(), eng.stream)
        elif m.likelihood.kind == "dgauss":
            self.lib.dgauss_nll_bwd(eng.dt, B, R, R, Cx, params.cv(), xin.cv(), self.coef.data_ptr(), 0, gparams.cv(), eng.stream)
        else:
            self.lib.dmol_nll_bwd(eng.dt, B, R, R, params.cv(), xin.cv(), self.coef.data_ptr(), 0, gparams.cv(), eng.stream)
        eng.backward()
        if self.acc_g is not None:
            eng.flat_axpy(eng.flat_g.data_ptr(), self.acc_g.data_ptr(), eng.flat_g.numel())
        return out3

    def _gbuf(self):
        return self.eng.flat_g if self.acc_g is None else self.acc_g

    def _allreduce(self, out3):
        """Average the flat gradient (and the reported scalars, whose NaN-ness feeds the skip predicate) over the
        data-parallel ranks: a few large buckets, not 800 small tensors (xGMI rings are per-link bound)."""
        if self.dp:
            t0 = None
            if self.time_comm:  # (serialized form: the whole exchange is exposed)
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
            dp.bucketed_allreduce_mean(self._gbuf(), self.bucket_elems, self.pg, extra=(out3,))
            if t0 is not None:
                t1.record()
                self._comm_events.append((t0, t1))

    # -- gradient all-reduce overlapped with the backward pass (north_star) -------------------------
    def _ranges_of(self, ids):
        """Contiguous [lo, hi) ranges of the flat gradient covered by the parameters with these ids."""
        eng = self.eng
        rs, cur = [], None
        for p in eng.params:
            o, k = eng.p_off[id(p)], p.numel()
            if id(p) in ids:
                if cur is not None and cur[1] == o:
                    cur[1] = o + k
                else:
                    cur = [o, o + k]
                    rs.append(cur)
            else:
                cur = None
        return [(a, b) for a, b in rs]

    def _early_launch(self):
        """Issue the all-reduce of the gradients that are already final (the decoder half, reduced by the background flush) on
        the communication stream, behind everything the main stream has done so far.  Returns the pending work handles."""
        main = torch.cuda.current_stream(self.eng.device)
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(self.eng.device)
        self.comm_stream.wait_stream(main)
        works = []
        g = self._gbuf()
        with torch.cuda.stream(self.comm_stream):
            for lo, hi in self.early_ranges:
                for o in range(lo, hi, self.bucket_elems):
                    works.append(torch.distributed.all_reduce(g[o:min(hi, o + self.bucket_elems)], group=self.pg, async_op=True))
        return works

    def _late_finish(self, out3, works):
        """All-reduce what the rest of the backward pass produced (plus the reported scalars), wait for both halves, average."""
        g = self._gbuf()
        t0 = None
        if self.time_comm:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        for lo, hi in self.late_ranges:
            for o in range(lo, hi, self.bucket_elems):
                works.append(torch.distributed.all_reduce(g[o:min(hi, o + self.bucket_elems)], group=self.pg, async_op=True))
        works.append(torch.distributed.all_reduce(out3, group=self.pg, async_op=True))
        for w in works:
            w.wait()
        g.mul_(1.0 / self.world)
        out3.mul_(1.0 / self.world)
        if t0 is not None:
            t1.record()
            self._comm_events.append((t0, t1))

    def exposed_comm_ms(self):
        """Mean time per step between the end of the backward pass and the end of the gradient exchange (what the all-reduce
        adds to the step; needs ``time_comm = True`` before the steps of interest)."""
        if not self._comm_events:
            return None
        torch.cuda.synchronize()
        v = [a.elapsed_time(b) for a, b in self._comm_events]
        self._comm_events = []
        return sum(v) / len(v)

    def _used_ranges(self):
        eng = self.eng
        rs, cur = [], None
        for p in eng.params:
            o, k = eng.p_off[id(p)], p.numel()
            used = id(p) in eng.pgrad_init and p.requires_grad
            if used:
                if cur is not None and cur[1] == o:
                    cur[1] = o + k
                else:
                    cur = [o, o + k]
                    rs.append(cur)
            else:
                cur = None
        return [(a, b) for a, b in rs]

    def _optim(self, out3):
        eng, a = self.eng, self.args
        st = eng.stream
        gbuf = self._gbuf()
        self.lib.sumsq_partial(gbuf.data_ptr(), gbuf.numel(), self.partial.data_ptr(), self.NORM_BLOCKS, st)
        self.lib.clip_decide(self.partial.data_ptr(), self.NORM_BLOCKS, out3.data_ptr(), float(a.grad_clip), float(a.grad_skip),
                             self.state.data_ptr(), st)
        if self.ranges is None:
            self.ranges = self._used_ranges()
        for (lo, hi) in self.ranges:
            q = _lib.AdamwArgs()
            q.p = eng.flat_p.data_ptr() + 4 * lo
            q.g = gbuf.data_ptr() + 4 * lo
            q.m = self.m.data_ptr() + 4 * lo
            q.v = self.v.data_ptr() + 4 * lo
            q.ema = self.ema_flat.data_ptr() + 4 * lo if self.ema_model is not None else None
            q.count = hi - lo
            q.lr, q.beta1, q.beta2, q.eps, q.wd = float(a.lr), float(a.betas[0]), float(a.betas[1]), 1e-8, float(a.wd)
            q.ema_beta, q.warmup_steps, q.ema_update_after = float(a.ema_rate), int(a.lr_warmup_steps), 100
            q.state_dev = self.state.data_ptr()
            self.lib.adamw_ema(C.byref(q), st)
        self.lib.step_commit(self.state.data_ptr(), st)
        if self.acc_g is not None:  # model.zero_grad() of trainer.py:87
            eng.flat_axpy(None, self.acc_g.data_ptr(), self.acc_g.numel(), alpha=0.0, accumulate=False)

    def _eager(self, x, pa, beta, do_step=True):
        self._coef_for(x, beta)
        overlap = self.dp_overlap and do_step and self.acc_g is None
        works, fired = [], []
        if overlap:
            def at_split():
                now = self._ranges_of(self.eng.early_final)
                if self.early_ranges is None:  # first step: which flat ranges are final here, and which come later
                    self.early_ranges = now
                # the ranges are fixed by the FIRST tape; another tape (other batch shape / drop_cond outcome) may cut at a
                # place where some of those gradients are still being written: such a step exchanges everything after the
                # backward pass instead (and its graph is captured without the cut)
                self._early_ok = _covers(now, self.early_ranges)
                if self._early_ok:
                    works.extend(self._early_launch())
                    fired.append(True)
            self._early_ok = True
            self.eng.on_split = at_split
        try:
            out3 = self._fwd_bwd(x, pa, beta)
        finally:
            self.eng.on_split = None
        self.eng.stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        if do_step:
            if overlap and not self._early_ok:
                self._allreduce(out3)
            elif overlap:
                if self.early_ranges is None:
                    self.early_ranges = []  # (the split mark was never reached: tiny model) -> everything is "late"
                elif not fired:
                    works.extend(self._early_launch())  # (a step without the background flush, e.g. the profiled one: exchange that half now)
                if self.late_ranges is None:
                    used = self._used_or_all_ranges()
                    self.late_ranges = _subtract_ranges(used, self.early_ranges)
                self._late_finish(out3, works)
            else:
                self._allreduce(out3)
            self._optim(out3)
            self._mark_weights_written()
        return out3

    def _used_or_all_ranges(self):
        return [(0, self._gbuf().numel())]

    def _mark_weights_written(self):
        """The fused AdamW / EMA kernel writes both flat parameter buffers through raw pointers, which torch's version
        counters do not see: tell the engines (model and EMA copy) that their weight images are stale, so the next
        inference call (validation on ``ema_model``, sampling, counterfactuals) re-images before it runs."""
        for mod in (self.model, self.ema_model):
            eng = None if mod is None else mod.__dict__.get("_eng")
            if eng is not None:
                eng.weights_dirty = True

    # -- public -------------------------------------------------------------------------------------------
    def step(self, x, pa):
        """One optimiser step on a batch already resident on the GPU.  Returns the device tensor [elbo, nll, kl]."""
        a = self.args
        self._loss_scale_check()
        self.it += 1
        beta = self.beta
        if getattr(a, "beta_warmup_steps", 0) > 0:
            beta = self.beta * linear_warmup(a.beta_warmup_steps)(self.it)
        do_step = (self.it - 1) % self.accu == 0
        if not self.use_graph:
            return self._eager(x, pa, beta, do_step)
        m = self.model
        drop = (1, 1)
        if m.cond_prior:  # host draw (shared across DP ranks through the common seed), one graph per outcome
            drop = type(m.decoder).drop_cond(m.decoder)
            m.decoder.__dict__["drop_cond"] = lambda d=drop: d
        # (beta is device data: the warm-up schedule replays the same graph; virtual and materialised parents are different graphs)
        key = (tuple(x.shape), x.dtype, drop, do_step, tuple(pa.shape), compact_parents(pa) is not None)
        ent = self.graphs.get(key)
        self._coef_for(x, beta)
        if ent is None:
            out = self._eager(x, pa, beta, do_step)  # eager warm-up: sizes the arena, builds the tables
            sx, spb = x.clone(), StaticParents(pa)
            sp = spb.t
            torch.cuda.synchronize()
            # NCCL inside a captured graph is avoided: under DP the step is graphs around eager all-reduces.  With overlap the
            # backward graph is cut where the decoder half of the gradient is final (engine.on_split): graph A | all-reduce of
            # that half on the communication stream | graph B (rest of the backward pass) | all-reduce of the rest | graph C.
            overlap = self.dp_overlap and do_step and self.acc_g is None and bool(self.early_ranges) and self._early_ok
            g1, g1b = torch.cuda.CUDAGraph(), None
            if overlap:
                g1b = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream(self.eng.device)
                cap.wait_stream(torch.cuda.current_stream(self.eng.device))
                fired = []

                def at_split():
                    g1.capture_end()
                    g1b.capture_begin(pool=g1.pool(), capture_error_mode=CAPTURE_MODE)
                    fired.append(True)

                with torch.cuda.stream(cap):
                    g1.capture_begin(capture_error_mode=CAPTURE_MODE)
                    self.eng.on_split = at_split
                    try:
                        so = self._fwd_bwd(sx, sp, beta)
                    finally:
                        self.eng.on_split = None
                    (g1b if fired else g1).capture_end()
                torch.cuda.current_stream(self.eng.device).wait_stream(cap)
                if not fired:
                    g1b = None
            else:
                with torch.cuda.graph(g1, capture_error_mode=CAPTURE_MODE):
                    so = self._fwd_bwd(sx, sp, beta)
                    if not self.dp and do_step:
                        self.eng.stream = torch.cuda.current_stream(self.eng.device).cuda_stream
                        self._optim(so)
            g2 = None
            if self.dp and do_step:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode=CAPTURE_MODE):
                    self.eng.stream = torch.cuda.current_stream(self.eng.device).cuda_stream
                    self._optim(so)
            self.graphs[key] = (g1, g2, sx, spb, so, g1b)
            return out
        g1, g2, sx, spb, so, g1b = ent
        if sx.data_ptr() != x.data_ptr():
            sx.copy_(x, non_blocking=True)
            spb.load(pa)
        g1.replay()
        if g1b is not None:
            works = self._early_launch()
            g1b.replay()
            self._late_finish(so, works)
            g2.replay()
        elif g2 is not None:
            self._allreduce(so)
            g2.replay()
        if do_step:
            self._mark_weights_written()
        return so

    # -- checkpoint / resume (trainer.py:154-165, main.py:75-90) -----------------------------------------------
    def state_dict(self):
        """Optimiser state in ``torch.optim.AdamW.state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` /
        ``exp_avg_sq`` in ``model.parameters()`` order, one param group) so that a reference-side resume can load it,
        plus this harness's own counters under ``"cgen"``."""
        eng, a = self.eng, self.args
        st = self.state.cpu()
        steps = float(st[5])
        state = {}
        for i, p in enumerate(eng.params):
            o, k = eng.p_off[id(p)], p.numel()
            state[i] = {"step": torch.tensor(steps), "exp_avg": self.m[o:o + k].view(p.shape).clone(),
                        "exp_avg_sq": self.v[o:o + k].view(p.shape).clone()}
        group = {"lr": float(a.lr) * linear_warmup(int(a.lr_warmup_steps))(int(steps)), "betas": tuple(float(b) for b in a.betas),
                 "eps": 1e-8, "weight_decay": float(a.wd), "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "initial_lr": float(a.lr), "params": list(range(len(eng.params)))}
        return {"state": state, "param_groups": [group],
                "cgen": {"it": self.it, "opt_steps": int(steps), "n_skipped": int(st[4]),
                         "loss_scale_shift": int(self.eng.loss_scale_shift), "overflow_backoffs": int(self.overflow_backoffs),
                         "ls_growth_interval": int(self.ls_growth_interval),
                         "acc_g": None if self.acc_g is None else self.acc_g.clone()}}

    def scheduler_state_dict(self):
        """``LambdaLR.state_dict()`` of the reference's warm-up scheduler (train_setup.py:49-51): one step per successful
        optimiser step."""
        steps = int(self.state[5].item())
        a = self.args
        lr = float(a.lr) * linear_warmup(int(a.lr_warmup_steps))(steps)
        return {"base_lrs": [float(a.lr)], "last_epoch": steps, "_step_count": steps + 1, "_get_lr_called_within_step": False,
                "_last_lr": [lr], "lr_lambdas": [None]}

    def load_state_dict(self, sd):
        """Inverse of :meth:`state_dict`; also accepts a plain ``torch.optim.AdamW`` state dict saved by the reference
        (the step count is then the parameters' common ``step``)."""
        eng = self.eng
        steps = None
        for i, p in enumerate(eng.params):
            ent = sd["state"].get(i)
            if ent is None:
                continue
            o, k = eng.p_off[id(p)], p.numel()
            self.m[o:o + k].copy_(ent["exp_avg"].reshape(-1))
            self.v[o:o + k].copy_(ent["exp_avg_sq"].reshape(-1))
            steps = float(ent["step"])
        extra = sd.get("cgen") or {}
        if "opt_steps" in extra:
            steps = float(extra["opt_steps"])
        st = torch.zeros(8)
        st[5] = 0.0 if steps is None else steps
        st[4] = float(extra.get("n_skipped", 0))
        self.state.copy_(st)
        self.it = int(extra.get("it", st[5]))
        self._ls_checked_it, self._clean_checks, self._ls_last_growth_it = self.it, 0, None
        self._ls_nonfinite_seen = 0.0  # (state[6] restarts at zero with the state vector)
        self.overflow_backoffs = int(extra.get("overflow_backoffs", 0))
        self.ls_growth_interval = int(extra.get("ls_growth_interval", self.ls_growth_interval))
        shift = max(self.LS_SHIFT_MIN, min(0, int(extra.get("loss_scale_shift", 0))))
        if shift != self.eng.loss_scale_shift:
            self._rescale(shift - self.eng.loss_scale_shift)
        if self.acc_g is not None and extra.get("acc_g") is not None:
            self.acc_g.copy_(extra["acc_g"])

    def _loss_scale_check(self):
        """Every `ls_check_interval` iterations (f16 engine only): one host read of the device step state.  A step since the
        previous check was dropped for a NON-FINITE gradient norm => an activation gradient left binary16's range: halve the scale (captured graphs
        and coefficient tables are rebuilt at this step) instead of dropping every later step as well.  After
        `ls_growth_interval` clean checks the scale is doubled back towards the rule's value; a growth that overflows again
        within one interval doubles that interval (no flapping).  The shift is clamped to [LS_SHIFT_MIN, 0]."""
        if self.eng.dtype_name == "f32" or self.it == 0 or self.it % self.ls_check_interval or self._ls_checked_it == self.it:
            return
        self._ls_checked_it = self.it
        s = self.state.cpu().tolist()
        # decided from a COUNTER, not from a sample of the last step: state[6] counts the steps dropped for a non-finite
        # norm; any of the ls_check_interval steps since the previous check having overflowed backs the scale off, and a check only
        # counts as clean when none did
        overflow = s[6] > self._ls_nonfinite_seen
        self._ls_nonfinite_seen = s[6]
        if overflow:
            if self.eng.loss_scale_shift > self.LS_SHIFT_MIN:
                self._rescale(-1)
                self.overflow_backoffs += 1
            if self._ls_last_growth_it is not None and self.it - self._ls_last_growth_it <= self.ls_growth_interval * self.ls_check_interval:
                self.ls_growth_interval = min(4096, 2 * self.ls_growth_interval)
            self._ls_last_growth_it = None
            self._clean_checks = 0
        else:
            self._clean_checks += 1
            if self.eng.loss_scale_shift < 0 and self._clean_checks >= self.ls_growth_interval:
                self._rescale(+1)
                self._clean_checks = 0
                self._ls_last_growth_it = self.it

    def stats(self):
        """Host read of the device-side step state (one sync; call every N steps, not every step).  Read-only: the f16
        loss-scale back-off lives in step() (`_loss_scale_check`); `overflow_backoffs` counts the halvings so far,
        `loss_scale_shift` is the current log2 offset from the rule's value."""
        s = self.state.cpu().tolist()
        out = dict(grad_norm=s[1], clip_coef=s[2], skipped_last=bool(s[3]), n_skipped=int(s[4]), opt_steps=int(s[5]))
        if self.eng.dtype_name != "f32":
            out.update(loss_scale=self.eng.loss_scale, loss_scale_shift=self.eng.loss_scale_shift, overflow_backoffs=self.overflow_backoffs)
        return out

    def _rescale(self, d):
        """Move the loss-scale back-off by `d` powers of two: everything that baked the old scale in is dropped (the captured
        step graphs, the seed coefficients, the engine's reduce tables are keyed by the scale)."""
        self.eng.loss_scale_shift = max(self.LS_SHIFT_MIN, min(0, self.eng.loss_scale_shift + d))
        torch.cuda.synchronize()
        self.graphs.clear()
        for ent in self.coefs.values():
            ent[1] = None
