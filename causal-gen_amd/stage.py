"""Host side of the per-image stage interpreter (csrc/stage.hip; include/cgen_hip.h "per-image stage interpreter").

The engine issues one fused launch per op through ``eng.lib.<entry point>(..., stream)``.  ``StagedLib`` is a proxy over the
ctypes library that sits exactly there: an op at a STAGED resolution (``Engine.stage_res``: the low resolutions of the
hierarchy, where a launch per conv is a pure latency chain) that the kernel serves is not launched but appended to the
engine's pending op list; any other call -- and every stream fork / join / event of the engine -- first flushes that list as
ONE launch of ``cgen_stage_run`` (one workgroup per image walks the list), then runs as before.  Program order on a stream is
therefore what it always was; vae.py / train.py do not know the difference.

The packed op table is planned once per distinct list (``cgen_stage_plan``) and kept on the device: the arena hands out the same
addresses every step, so a captured step replays it.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (F16, NULL_VIEW, ST_AVGPOOL_BWD, ST_AVGPOOL_FWD, ST_AXPBY, ST_BCAST, ST_CONV, ST_REPARAM_BWD, ST_REPARAM_FWD,
                   ST_UPSAMPLE_BWD, ST_UPSAMPLE_FWD, StageElemArgs, StageReparamArgs, StageReparamBwdArgs)

# entry points that answer on the host and launch nothing
_PURE = {"version", "last_error", "block3_supported", "block3_pair_supported", "conv2d_pair_supported", "stem_conv_supported", "stage_accepts", "stage_plan",
         "reparam_kl_chunks", "like_chunks", "conv2d_wgrad_plan", "conv2d_wgrad_batch_plan"}


def _elem(kind, dtype, n, h, w, out, inp=NULL_VIEW, d=0, hi=0, wi=0, accumulate=0, c_from=0, alpha=0.0, beta=0.0, src=None):
    a = StageElemArgs()
    a.dtype, a.n, a.h, a.w, a.d, a.hi, a.wi, a.accumulate, a.c_from = dtype, n, h, w, d, hi, wi, accumulate, c_from
    a.alpha, a.beta, a.src, a.inp, a.out = alpha, beta, src, inp, out
    return kind, a, n, max(h * max(d, 1), hi)


def _h_conv2d(args):
    a = args[0]._obj  # (C.byref(ConvArgs), stream)
    return ST_CONV, a, a.n, a.h


def _h_avgpool_fwd(args):
    dtype, n, ho, wo, d, inp, out, _ = args
    return _elem(ST_AVGPOOL_FWD, dtype, n, ho, wo, out, inp, d=d)


def _h_avgpool_bwd(args):
    dtype, n, ho, wo, d, gout, gin, acc, _ = args
    return _elem(ST_AVGPOOL_BWD, dtype, n, ho, wo, gin, gout, d=d, accumulate=acc)


def _h_upsample_fwd(args):
    dtype, n, hi, wi, ho, wo, inp, bias, out, _ = args
    return _elem(ST_UPSAMPLE_FWD, dtype, n, ho, wo, out, inp, hi=hi, wi=wi, src=bias)


def _h_upsample_bwd(args):
    dtype, n, hi, wi, ho, wo, gout, gin, acc, _ = args
    return _elem(ST_UPSAMPLE_BWD, dtype, n, ho, wo, gin, gout, hi=hi, wi=wi, accumulate=acc)


def _h_batch_broadcast(args):
    dtype, n, h, w, src, out, _ = args
    return _elem(ST_BCAST, dtype, n, h, w, out, src=src)


def _h_axpby(args):
    dtype, n, h, w, inp, out, alpha, beta, c_from, acc, _ = args
    return _elem(ST_AXPBY, dtype, n, h, w, out, inp, alpha=alpha, beta=beta, c_from=min(int(c_from), (1 << 31) - 1), accumulate=acc)


def _h_reparam_kl_fwd(args):
    dtype, n, h, w, c, q_loc, q_ls, p_loc, p_ls, eps_in, rng, stream_id, logt, z, eps_out, kl_part, kl_stride, _ = args
    if eps_out.p:
        return None
    a = StageReparamArgs()
    a.dtype, a.n, a.h, a.w, a.c, a.kl_stride, a.stream_id, a.logt = dtype, n, h, w, c, kl_stride, stream_id, logt
    a.q_loc, a.q_ls, a.p_loc, a.p_ls, a.eps_in, a.z, a.rng, a.kl_part = q_loc, q_ls, p_loc, p_ls, eps_in, z, rng, kl_part
    return ST_REPARAM_FWD, a, n, h


def _reparam_bwd(args, rider):
    (dtype, n, h, w, c, q_loc, q_ls, p_loc, p_ls, z, logt, gz, coef, coef_stride, chan, gql, gqs, gpl, gps, acc_q, acc_p) = args[:21]
    a = StageReparamBwdArgs()
    a.dtype, a.n, a.h, a.w, a.c, a.coef_stride, a.acc_q, a.acc_p, a.logt = dtype, n, h, w, c, coef_stride, acc_q, acc_p, logt
    a.q_loc, a.q_ls, a.p_loc, a.p_ls, a.z, a.gz = q_loc, q_ls, p_loc, p_ls, z, gz
    a.g_q_loc, a.g_q_ls, a.g_p_loc, a.g_p_ls = gql, gqs, gpl, gps
    a.kl_coef_dev, a.kl_chan_scale = coef, chan
    if rider:
        a.ride_src, a.ride_dst, a.ride_acc = args[21], args[22], args[23]
    else:
        a.ride_src = a.ride_dst = NULL_VIEW
    return ST_REPARAM_BWD, a, n, h


_HANDLERS = {
    "conv2d": _h_conv2d, "avgpool_fwd": _h_avgpool_fwd, "avgpool_bwd": _h_avgpool_bwd, "upsample_fwd": _h_upsample_fwd,
    "upsample_bwd": _h_upsample_bwd, "batch_broadcast": _h_batch_broadcast, "axpby": _h_axpby, "reparam_kl_fwd": _h_reparam_kl_fwd,
    "reparam_kl_bwd": lambda a: _reparam_bwd(a, False), "reparam_kl_bwd_rider": lambda a: _reparam_bwd(a, True),
}


class StagedLib:
    """Proxy over the ctypes library: stage-eligible launches are deferred, everything else flushes the pending list first."""

    def __init__(self, lib, eng):
        self.__dict__["_lib"] = lib
        self.__dict__["_eng"] = eng

    def __getattr__(self, name):
        lib, eng = self.__dict__["_lib"], self.__dict__["_eng"]
        fn = getattr(lib, name)
        if name in _PURE or not callable(fn):
            self.__dict__[name] = fn
            return fn
        handler = _HANDLERS.get(name)
        if handler is None:
            def call(*a):
                if eng._stage_ops:
                    eng.stage_flush()
                return fn(*a)
        else:
            def call(*a):
                if eng.stage_active and eng._stage_try(handler, a, fn):
                    return None
                if eng._stage_ops:
                    eng.stage_flush()
                return fn(*a)
        call.__name__ = name
        self.__dict__[name] = call
        return call


class StageMixin:
    """Engine side: policy, the pending list, the flush."""

    def _stage_init(self, rawlib):
        self._rawlib = rawlib
        # OFF by default: correct (tests/test_gpu_stage.py) and 35-95 % fewer launches, but on MI355X its conv body is not yet faster
        # than the tuned launch-per-op kernels it replaces (DESIGN 3.8 has the per-phase stamps: the K loop waits out one L2
        # round trip per K-step, ~1 us, where the MFMAs of a step take 0.05 us).  CGEN_STAGE=1 turns it on.
        self.stage_enabled = os.environ.get("CGEN_STAGE", "0") != "0"
        self.stage_res = frozenset()   # resolutions (image side) whose ops are staged; set by the model (HVAE.engine)
        self._stage_ops = []           # [(kind, args struct)] of the pending list
        self._stage_stream = None
        self._stage_n = None
        self._stage_tabs = {}
        self._stage_flops = {}
        self._stage_deferred = False
        self.stage_launches = 0
        self.stage_ops_total = 0
        self.stage_capture_misses = 0

    @property
    def stage_active(self):
        return self.stage_enabled and self.dt == F16 and bool(self.stage_res)

    def stage_covers(self, res):
        """True when the ops of a layer at this resolution go into stage lists (the forward pass then keeps them on ONE stream:
        a fork would only cut the list)."""
        return self.stage_active and res in self.stage_res

    def _stage_try(self, handler, a, fn):
        got = handler(a)
        if got is None:
            return False
        kind, args, n, res = got
        if res not in self.stage_res or args.dtype != F16:
            return False
        stream = a[-1]
        if not self._rawlib.stage_accepts(kind, C.addressof(args)):
            return False
        if self._stage_ops and (stream != self._stage_stream or n != self._stage_n):
            self.stage_flush()
        self._stage_stream, self._stage_n = stream, n
        self._stage_ops.append((kind, args, fn, a))
        self._stage_deferred = True
        return True

    def stage_flush(self):
        """Launch the pending op list (one workgroup per image) on the stream its ops were issued on."""
        ops = self._stage_ops
        if not ops:
            return
        self._stage_ops = []
        n = len(ops)
        key = b"".join(bytes(C.c_int32(o[0])) + bytes(o[1]) for o in ops)
        ent = self._stage_tabs.get(key)
        if ent is None and torch.cuda.is_current_stream_capturing():
            # A list first seen inside a stream capture (an operand allocated by torch inside the captured region has another
            # address than in the eager warm-up pass): its table cannot be uploaded here.  The ops run as the stand-alone
            # launches they would have been -- same results, launch-per-op speed for this list.
            self.stage_capture_misses += 1
            for _, _, fn, call_args in ops:
                fn(*call_args)
            self._stage_flops = {}
            return
        ops = [(o[0], o[1]) for o in ops]
        if ent is None:
            kinds = (C.c_int32 * n)(*[k for k, _ in ops])
            ptrs = (C.c_void_p * n)(*[C.addressof(a) for _, a in ops])
            nbytes, lds = C.c_int64(0), C.c_int32(0)
            self._rawlib.stage_plan(kinds, ptrs, n, None, 0, C.byref(nbytes), C.byref(lds))
            host = (C.c_char * nbytes.value)()
            self._rawlib.stage_plan(kinds, ptrs, n, host, nbytes.value, C.byref(nbytes), C.byref(lds))
            # synchronous upload, once per distinct list: like the weight-gradient batch tables, every list of a captured step
            # has been planned by the eager warm-up step that precedes the capture (same addresses: the arena is deterministic)
            dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
            host = None
            # never evicted: a captured hipGraph (TrainStep.graphs, GraphedCounterfactual) has this table's ADDRESS baked into its
            # cgen_stage_run node -- dropping the tensor would let the caching allocator hand the memory out again and a later
            # replay would walk a foreign op table.  A table is a few KB; a model has a few hundred distinct lists.
            ent = self._stage_tabs[key] = (dev, host, lds.value)
        dev, _, lds = ent
        if os.environ.get("CGEN_STAGE_DEBUG"):
            import sys
            desc = []
            for k, a in ops:
                if k == ST_CONV:
                    desc.append("conv%dx%d[%s->%d %dx%d act%d dact%d%s%s%s]" % (a.ks, a.ks, "+".join(str(a.seg[i].c) for i in range(a.nseg)), a.out.c, a.h, a.w,
                                                                          a.act, a.dact, " aux" if a.aux.p else "", " r1" if a.res1.p else "", " r2" if a.res2.p else ""))
                else:
                    desc.append("k%d[%dx%d]" % (k, a.h, a.w))
            print("stage launch: n=%d lds=%d stream=%x ops=%d: %s" % (self._stage_n, lds, self._stage_stream or 0, n, " ".join(desc)), file=sys.stderr, flush=True)
        ev = None
        if self.prof is not None and self._stage_flops:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._rawlib.stage_run(dev.data_ptr(), n, self._stage_n, lds, self._stage_stream)
        if os.environ.get("CGEN_STAGE_DEBUG") == "2":
            torch.cuda.synchronize()
        if ev is not None:
            ev[1].record()
            kind = max(self._stage_flops, key=lambda k: self._stage_flops[k])
            e = self.prof.setdefault((kind, 0, 0, 0, -1), [0.0, [], 0])
            e[0] += sum(self._stage_flops.values())
            e[1].append(ev)
            e[2] += 1
        self._stage_flops = {}
        self.launches += 1 - n  # (every deferred op was counted as a launch when it was issued)
        self.stage_launches += 1
        self.stage_ops_total += n
