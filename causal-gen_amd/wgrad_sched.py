"""Weight-gradient scheduling of the engine (a mixin of `Engine`): every weight gradient of a backward pass is a leaf, so they are
deferred, planned once as horizontally batched launches (`cgen_conv2d_wgrad_batch_plan / _run`: one launch walks the record list
of all pending problems), flushed in the background of the backward chain on a side stream with a capped grid, and their split-K
partials reduced by multi-tensor launches (`cgen_wgrad_reduce`) in rounds with at most one event per conv site.  DESIGN.md 3.3.
(reference: what `loss.backward()` does for the conv weights in trainer.py:64-67)"""
import ctypes as C
import os

import torch

from . import _lib


class WgradMixin:
    @staticmethod
    def _needs_wgrad(site):
        """The weight-and-bias gradient kernel runs when EITHER needs a gradient (x_like='shared_*' freezes the weight of
        likelihood.x_logscale but trains its bias, vae.py:342-344); only the parts that require grad become visible."""
        b = site.conv.bias
        return site.conv.weight.requires_grad or (b is not None and b.requires_grad)

    def _wgrad(self, site, segs, act, g):
        x0 = segs[0]
        a = _lib.WgradArgs()
        gn, gh, gw, vw = self._geom(site.ks, list(segs) + [g])
        a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act = self.dt, gn, gh, gw, site.ks, len(segs), act
        for k, s in enumerate(segs):
            a.seg[k] = vw(s)
        a.gout = vw(g)
        use = sum(1 for e in self._wg_events if e[0] is site)
        # (the plan -- kernel, split count, partial LAYOUT -- also depends on how the views are laid out: strides and 16-byte alignment
        #  decide between the streaming, tiled and generic kernels.  They are part of the key, so a differently laid-out view of the
        #  same site can never be reduced with a stale layout)
        lay = tuple((v.sn, v.sh, v.sw, v.c, v.p % 16 if v.p else 0) for v in [a.seg[k] for k in range(len(segs))] + [a.gout])
        key = (site.index, use, x0.n, x0.h, x0.w, lay)
        ent = self._partials.get(key)
        nw = site.co * site.taps * site.ci
        if ent is None:
            kind = C.c_int32(0)
            nsplit = self.lib.conv2d_wgrad_plan(C.byref(a), C.byref(kind))
            # (kind 3: the streaming kernel leaves its partials as [ci][flipped tap][co]; cgen_wgrad_reduce is told so)
            ent = (torch.empty(nsplit * (nw + site.co), dtype=torch.float32, device=self.device), nsplit, 1 if kind.value == 3 else 0)
            self._partials[key] = ent  # (reduce tables are keyed by these keys: a new key cannot invalidate an existing table)
        buf, nsplit, _layout = ent
        a.nsplit = nsplit
        a.partial_w = buf.data_ptr()
        a.partial_b = buf.data_ptr() + 4 * nsplit * nw if site.conv.bias is not None else None
        if self._defer_wgrad():
            cost = 2.0 * site.ci * site.taps * site.co * x0.n * x0.h * x0.w
            self._wg_deferred.append((a, cost))
            self._wg_cum += cost
            self._wgrad_marks()
        else:
            self._timed("conv_wgrad", site, x0, lambda: self.lib.conv2d_wgrad(C.byref(a), self.stream))
        self._wg_events.append((site, key, nsplit))

    def _wgrad_marks(self):
        """Background-flush and data-parallel split marks of the weight-gradient work deferred so far (see _wgrad)."""
        k = self._wg_nflush  # cumulative-cost marks, fractions of the pass's total
        if self._in_side:  # (a side-strand op of backward(): the mark is looked at again by the next main-strand weight gradient)
            return
        if (self.wgrad_batch and self.prof is None and k < len(self.wgrad_flush_frac) and self._wg_total > 0
                and self._wg_cum >= self.wgrad_flush_frac[k] * self._wg_total):
            self._wg_nflush += 1
            self._launch_batched_wgrads(background=True)
        if (self.on_split is not None and not self._split_done and self._wg_nflush >= max(1, len(self.wgrad_flush_frac)) and self._wg_total > 0
                and self._wg_cum >= self.split_frac * self._wg_total):
            self._split_done = True
            if self._wg_forked:  # join the background flush + its reduction: by now it has long finished (no stall)
                main = torch.cuda.current_stream(self.device)
                for st in self._wg_pool:
                    main.wait_stream(st)
                self._wg_forked = False
            final = set()
            for st_, _, _ in self._wg_events[:self._wg_reduced]:
                final.add(id(st_.conv.weight))
                if st_.conv.bias is not None:
                    final.add(id(st_.conv.bias))
            self.early_final = final
            self.on_split()

    def _launch_deferred_wgrads(self, final=True):
        main = torch.cuda.current_stream(self.device)
        if self.wgrad_batch:
            if self._wg_deferred:
                self._launch_batched_wgrads()
            if final and self._wg_forked:
                for st in self._wg_pool:
                    main.wait_stream(st)
                self._wg_forked = False
            return
        if self._wg_deferred:
            k = min(self.wgrad_streams, len(self._wg_deferred))
            while len(self._wg_pool) < k:
                self._wg_pool.append(torch.cuda.Stream(self.device))
            streams = self._wg_pool[:k]
            for st in streams:
                st.wait_stream(main)  # fork: everything enqueued so far is visible
            self._wg_forked = True
            load = [0.0] * k
            for a, cost in sorted(self._wg_deferred, key=lambda e: -e[1]):  # longest first onto the least loaded stream
                i = load.index(min(load))
                load[i] += cost + 2.0e8  # + a fixed per-launch cost
                self.lib.conv2d_wgrad(C.byref(a), streams[i].cuda_stream)
                self.launches += 1
            self._wg_deferred = []
        if final and self._wg_forked:
            for st in self._wg_pool:
                main.wait_stream(st)
            self._wg_forked = False

    def _launch_batched_wgrads(self, background=False):
        """All deferred weight-gradient problems in a handful of launches.  The packed problem table is planned once per
        distinct set of launch arguments (addresses are stable: the arena is deterministic) and kept on the device."""
        args = [a for a, _ in self._wg_deferred]
        n = len(args)
        if os.environ.get("CGEN_WG_DEBUG"):
            tot, line = 0.0, []
            for a, c in self._wg_deferred:
                tot += c
                line.append("%dx%d:%.0f" % (a.h, a.ks, tot / 1e9))
            print("wgrad batch (bg=%s): %d problems, %.0f GF: %s" % (background, n, tot / 1e9, " ".join(line)), flush=True)
            print("   arena chunks %s; partial buffers %s; flat_g %s" % (
                ["%x-%x" % (c.data_ptr(), c.data_ptr() + c.numel()) for c in self.arena.chunks],
                ["%x-%x" % (a.partial_w, a.partial_w + 1) for a in args][:4],
                "%x-%x" % (self.flat_g.data_ptr(), self.flat_g.data_ptr() + 4 * self.flat_g.numel())), flush=True)
            for key, (buf, ns, _lay) in list(self._partials.items())[-4:]:
                print("   partial %s: %x-%x nsplit %d" % (key, buf.data_ptr(), buf.data_ptr() + 4 * buf.numel(), ns), flush=True)
        arr = (_lib.WgradArgs * n)(*args)
        key = bytes(arr)
        ent = self._wg_batches.get(key)
        if ent is None:
            lib = self.lib
            nbytes, nl = C.c_int64(0), C.c_int32(0)
            elig = (C.c_int32 * n)()
            lib.conv2d_wgrad_batch_plan(arr, n, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
            host = (C.c_char * max(nbytes.value, 1))()
            launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
            lib.conv2d_wgrad_batch_plan(arr, n, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
            blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device) if nbytes.value else None
            rest = [i for i in range(n) if not elig[i]]
            if len(self._wg_batches) > 64:
                self._wg_batches.clear()
            ent = self._wg_batches[key] = (blob, launches, nl.value, rest)
            if os.environ.get("CGEN_WG_DEBUG"):
                print("  launches: " + " ".join("<%d,ks%d>lds%dK:%dblk" % (launches[i].ncf, launches[i].ks, launches[i].lds_bytes // 1024, launches[i].nblocks) for i in range(nl.value)), flush=True)
        blob, launches, nl, rest = ent
        ev = None
        if self.prof is not None:  # the packed launches are timed as ONE class entry (per-problem times do not exist)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if background:
            main = torch.cuda.current_stream(self.device)
            if not self._wg_pool:
                self._wg_pool.append(torch.cuda.Stream(self.device))
            side = self._wg_pool[0]
            snap = None
            if os.environ.get("CGEN_WG_STRAYCHECK"):  # debugging: does the flushed batch write anywhere inside the arena?
                torch.cuda.synchronize()
                snap = [c.clone() for c in self.arena.chunks]
            # problems the packed kernel does not serve (f32, < 3x3 images) stay in line on the main stream: only the
            # packed launch goes to the background (the generic kernel on the side stream was the one configuration in
            # which tools/fuzz_conv.py produced a GPU memory fault; it is also a negligible share of the bf16 work)
            for i in rest:
                self.lib.conv2d_wgrad(C.byref(args[i]), self.stream)
                self.launches += 1
            if os.environ.get("CGEN_WGRAD_BG_SERIAL"):  # debugging: same launches, but in line on the main stream
                side = main
            else:
                side.wait_stream(main)
                if getattr(self, "_bw_forked", False):  # problems queued by side-strand ops read what that strand wrote
                    side.wait_stream(self._fwd_side)
                self._wg_forked = True
            if blob is not None and nl:
                self.lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl, self.wgrad_bg_wgs, side.cuda_stream)
                self.launches += nl
            self._wg_deferred = []
            if self.wgrad_bg_reduce:
                self._reduce_events(side.cuda_stream)  # these partials are final: their reduction leaves the critical path too
            if snap is not None:
                torch.cuda.synchronize()
                for k, (c0, c1) in enumerate(zip(snap, self.arena.chunks)):
                    d = (c0 != c1).nonzero().flatten()
                    print("straycheck chunk %d: %d arena bytes changed by the flushed batch (arena offset now %d of chunk %d)%s" % (
                        k, d.numel(), self.arena.off, self.arena.ci, "" if not d.numel() else " first %d last %d" % (int(d[0]), int(d[-1]))), flush=True)
            return
        if blob is not None and nl:
            self.lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl, 0, self.stream)
            self.launches += nl
        if len(rest) >= 4 and self.wgrad_streams > 1 and ev is None:
            # problems the packed kernel does not serve (f32, < 5x5 images): independent launches on the stream pool
            main = torch.cuda.current_stream(self.device)
            k = min(self.wgrad_streams, len(rest))
            while len(self._wg_pool) < k:
                self._wg_pool.append(torch.cuda.Stream(self.device))
            costs = [c for _, c in self._wg_deferred]
            load = [0.0] * k
            for st in self._wg_pool[:k]:
                st.wait_stream(main)
            for i in sorted(rest, key=lambda i: -costs[i]):
                j = load.index(min(load))
                load[j] += costs[i] + 2.0e8
                self.lib.conv2d_wgrad(C.byref(args[i]), self._wg_pool[j].cuda_stream)
                self.launches += 1
            for st in self._wg_pool[:k]:
                main.wait_stream(st)
        else:
            for i in rest:
                self.lib.conv2d_wgrad(C.byref(args[i]), self.stream)
                self.launches += 1
        if ev is not None:
            ev[1].record()
            ent2 = self.prof.setdefault(("conv_wgrad", 0, 0, 0, 0), [0.0, [], 0])
            ent2[0] += sum(c for _, c in self._wg_deferred)
            ent2[1].append(ev)
            ent2[2] += nl + len(rest)
        self._wg_deferred = []

    def _reduce_wgrads(self):
        self._launch_deferred_wgrads()
        self._reduce_events(self.stream)

    def _reduce_events(self, stream):
        """Split-K partials -> flat OIHW gradients for every weight-gradient event not reduced yet (one multi-tensor launch).
        Called on the side stream right after a background flush (those partials are final) and at the end of the pass."""
        events = self._wg_events[self._wg_reduced:]
        if not events:
            return
        self._wg_reduced = len(self._wg_events)
        # One multi-tensor launch must not hold two events of the SAME site (DSCM.forward's tape runs every conv up to three
        # times): their blocks would overwrite / accumulate the same gradient tensor concurrently.  Events are dealt into
        # rounds with at most one event per site; the rounds run back to back on the stream.
        rounds, nth = [], {}
        for ev in events:
            k = nth.get(ev[0].index, 0)
            nth[ev[0].index] = k + 1
            while len(rounds) <= k:
                rounds.append([])
            rounds[k].append(ev)
        for evs in rounds:
            self._reduce_round(evs, stream)

    def _reduce_round(self, events, stream):
        flags = []
        for site, _, _ in events:  # a site used before in this pass accumulates
            flags.append(site.index in self._wg_seen)
            self._wg_seen.add(site.index)
        sig = (tuple(k for _, k, _ in events), tuple(flags), self.loss_scale)
        tab = self._red_tabs.get(sig)
        if tab is None:
            descs, csite, cidx = [], [], []
            for (site, key, nsplit), acc in zip(events, flags):
                buf = self._partials[key][0]
                nw = site.co * site.taps * site.ci
                d = _lib.WredDesc()
                d.partial_w = buf.data_ptr()
                d.partial_b = buf.data_ptr() + 4 * nsplit * nw if site.conv.bias is not None else None
                d.grad_w = self.param_grad_ptr(site.conv.weight)
                d.grad_b = self.param_grad_ptr(site.conv.bias) if site.conv.bias is not None else None
                d.co, d.ci_total, d.ks, d.nsplit = site.co, site.ci, site.ks, nsplit
                d.layout = self._partials[key][2]
                d.accumulate = 1 if acc else 0
                d.unscale = 1.0 / self.loss_scale
                d.numel = nw + (site.co if site.conv.bias is not None else 0)
                descs.append(d)
            for i, d in enumerate(descs):
                nch = (d.numel + self.CHUNK - 1) // self.CHUNK
                csite += [i] * nch
                cidx += list(range(nch))
            arr = (_lib.WredDesc * len(descs))(*descs)
            tab = (self._to_dev(bytes(arr)), torch.tensor(csite, dtype=torch.int32, device=self.device),
                   torch.tensor(cidx, dtype=torch.int32, device=self.device), len(csite))
            self._red_tabs[sig] = tab
        d, cs, ci, n = tab
        self.lib.wgrad_reduce(d.data_ptr(), cs.data_ptr(), ci.data_ptr(), n, stream)
        self.launches += 1
        for site, _, _ in events:
            if site.conv.weight.requires_grad:
                self.pgrad_init.add(id(site.conv.weight))
            if site.conv.bias is not None and site.conv.bias.requires_grad:
                self.pgrad_init.add(id(site.conv.bias))

