"""ctypes binding of libcgen_hip.so (include/cgen_hip.h).  Loading never needs a GPU; calling does."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGEN_LIB") or os.path.join(_HERE, "libcgen_hip.so")  # (CGEN_LIB: another build of the same ABI, for A/B runs)

F32, F16, F32S = 0, 1, 2
DMOL_LOW_BIT = 0x100  # OR-ed into the dtype of cgen_dmol_nll_fwd/bwd: the reference's low_bit=True branch (dmol.py:52-60)
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
UNARY_LEAKY_RELU, UNARY_CLAMP_MIN, UNARY_ADD = 3, 4, 5
MAX_SEG = 4


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sn", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64), ("c", C.c_int32),
                ("cpad", C.c_int32)]


NULL_VIEW = View(None, 0, 0, 0, 0, 0)


class ConvArgs(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("ks", C.c_int32),
                ("nseg", C.c_int32), ("act", C.c_int32), ("dact", C.c_int32), ("seg", View * MAX_SEG),
                ("weight", C.c_void_p), ("bias", C.c_void_p), ("out", View), ("aux", View), ("res1", View), ("res2", View),
                ("out_rem", C.c_int64), ("res1_rem", C.c_int64)]


class Block3Out(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("out", View), ("aux", View), ("res1", View),
                ("out_rem", C.c_int64), ("res1_rem", C.c_int64)]


class Block3Args(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("nseg", C.c_int32),
                ("nout", C.c_int32), ("pre_act", C.c_int32), ("reserved", C.c_int32), ("seg", View * MAX_SEG),
                ("w_a", C.c_void_p), ("bias_a", C.c_void_p), ("mid", View), ("mid_aux", View), ("o", Block3Out * 2), ("w_a16", C.c_void_p)]


class Block4Args(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("nseg", C.c_int32),
                ("nout", C.c_int32), ("fwd", C.c_int32), ("b", C.c_int32), ("seg", View * MAX_SEG),
                ("wimg", C.c_void_p * 3), ("bias", C.c_void_p * 3), ("mid", View * 3), ("mid_aux", View * 3), ("o", Block3Out * 3)]


class WgradArgs(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("ks", C.c_int32),
                ("nseg", C.c_int32), ("act", C.c_int32), ("nsplit", C.c_int32), ("seg", View * MAX_SEG), ("gout", View),
                ("partial_w", C.c_void_p), ("partial_b", C.c_void_p)]


class WprepDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("co", C.c_int32), ("ci_total", C.c_int32), ("ks", C.c_int32),
                ("mode", C.c_int32), ("nseg", C.c_int32), ("seg_off", C.c_int32), ("seg_c", C.c_int32 * MAX_SEG),
                ("dtype", C.c_int32), ("rows_pad", C.c_int32), ("k_pad", C.c_int32), ("reserved", C.c_int32),
                ("numel", C.c_int64)]


class WredDesc(C.Structure):
    _fields_ = [("partial_w", C.c_void_p), ("partial_b", C.c_void_p), ("grad_w", C.c_void_p), ("grad_b", C.c_void_p),
                ("co", C.c_int32), ("ci_total", C.c_int32), ("ks", C.c_int32), ("nsplit", C.c_int32),
                ("accumulate", C.c_int32), ("unscale", C.c_float), ("layout", C.c_int32), ("reserved", C.c_int32),
                ("numel", C.c_int64)]


class AdamwArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("count", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("wd", C.c_float), ("ema_beta", C.c_float), ("warmup_steps", C.c_int32), ("ema_update_after", C.c_int32),
                ("state_dev", C.c_void_p)]


i32, i64, u32, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_void_p

# name -> argtypes (all return int unless listed in _RESTYPES).  Kept in lock-step with include/cgen_hip.h;
# tests/test_abi.py parses the header and checks every declared symbol is exported and bound here.
PROTOTYPES = {
    "cgen_version": [],
    "cgen_h16_format": [],
    "cgen_last_error": [],
    "cgen_conv2d": [C.POINTER(ConvArgs), vp],
    "cgen_conv2d_pair_supported": [C.POINTER(ConvArgs), C.POINTER(ConvArgs)],
    "cgen_conv2d_pair": [C.POINTER(ConvArgs), C.POINTER(ConvArgs), vp],
    "cgen_block3_supported": [C.POINTER(Block3Args)],
    "cgen_block3": [C.POINTER(Block3Args), vp],
    "cgen_block3_pair_supported": [C.POINTER(Block3Args), C.POINTER(Block3Args)],
    "cgen_block3_pair": [C.POINTER(Block3Args), C.POINTER(Block3Args), vp],
    "cgen_block4_supported": [C.POINTER(Block4Args)],
    "cgen_block4": [C.POINTER(Block4Args), vp],
    "cgen_block4_pair_supported": [C.POINTER(Block4Args), C.POINTER(Block4Args)],
    "cgen_block4_pair": [C.POINTER(Block4Args), C.POINTER(Block4Args), vp],
    "cgen_conv2d_wgrad_plan": [C.POINTER(WgradArgs), C.POINTER(i32)],
    "cgen_conv2d_wgrad_batch_plan": [vp, i32, vp, i64, vp, vp, i32, vp, vp],
    "cgen_conv2d_wgrad_batch_run": [vp, vp, i32, i32, vp],
    "cgen_conv2d_wgrad": [C.POINTER(WgradArgs), vp],
    "cgen_weight_prep": [vp, vp, vp, i32, vp],
    "cgen_wgrad_reduce": [vp, vp, vp, i32, vp],
    "cgen_avgpool_fwd": [i32, i32, i32, i32, i32, View, View, vp],
    "cgen_avgpool_bwd": [i32, i32, i32, i32, i32, View, View, i32, vp],
    "cgen_adaptive_avgpool_fwd": [i32, i32, i32, i32, i32, i32, View, View, vp],
    "cgen_adaptive_avgpool_bwd": [i32, i32, i32, i32, i32, i32, View, View, i32, vp],
    "cgen_upsample_fwd": [i32, i32, i32, i32, i32, i32, View, vp, View, vp],
    "cgen_upsample_bwd": [i32, i32, i32, i32, i32, i32, View, View, i32, vp],
    "cgen_batch_reduce": [i32, i32, i32, i32, View, vp, i32, f32, vp],
    "cgen_batch_broadcast": [i32, i32, i32, i32, vp, View, vp],
    "cgen_axpby": [i32, i32, i32, i32, View, View, f32, f32, i32, i32, vp],
    "cgen_nchw_to_nhwc": [i32, i32, i32, i32, i32, i32, vp, View, f32, f32, vp],
    "cgen_nhwc_to_nchw": [i32, i32, i32, i32, i32, View, vp, vp],
    "cgen_stem_conv_supported": [i32, i32, i32, i32],
    "cgen_stem_conv_fwd": [i32, i32, i32, i32, i32, i32, i32, View, vp, vp, View, vp],
    "cgen_im2col": [i32, i32, i32, i32, i32, View, View, vp],
    "cgen_reparam_kl_chunks": [i32, i32, i32],
    "cgen_reparam_kl_fwd": [i32, i32, i32, i32, i32, View, View, View, View, View, vp, u32, f32, View, View, vp, i32, vp],
    "cgen_reparam_kl_bwd": [i32, i32, i32, i32, i32, View, View, View, View, View, f32, View, vp, i32, vp, View, View, View,
                            View, i32, i32, vp],
    "cgen_reparam_kl_bwd_rider": [i32, i32, i32, i32, i32, View, View, View, View, View, f32, View, vp, i32, vp, View, View, View,
                                  View, i32, i32, View, View, i32, vp],
    "cgen_kl_channel_sums": [i32, i32, i32, i32, i32, View, View, View, View, f32, vp, i32, vp],
    "cgen_elbo_finalize_fb": [i32, vp, i32, f32, vp, i32, f32, f32, f32, vp, vp, vp, vp],
    "cgen_im2col_strided": [i32, i32, i32, i32, i32, i32, i32, i32, i32, View, View, vp],
    "cgen_col2im_strided": [i32, i32, i32, i32, i32, i32, i32, i32, i32, View, View, i32, vp],
    "cgen_unary_fwd": [i32, i32, f32, i32, i32, i32, i32, View, View, vp],
    "cgen_unary_bwd": [i32, i32, f32, i32, i32, i32, i32, View, View, View, i32, vp],
    "cgen_sample_gaussian": [i32, i32, i32, i32, i32, View, View, View, vp, u32, f32, View, vp],
    "cgen_gaussian_kl_map": [i64, vp, vp, vp, vp, vp, vp],
    "cgen_mediator_mix": [i32, i32, i32, i32, i32, View, View, View, View, View, f32, f32, f32, i32, View, vp],
    "cgen_like_chunks": [i32, i32],
    "cgen_dgauss_nll_fwd": [i32, i32, i32, i32, i32, View, View, vp, vp],
    "cgen_dgauss_nll_bwd": [i32, i32, i32, i32, i32, View, View, vp, i32, View, vp],
    "cgen_dgauss_params": [i32, i32, i32, i32, i32, View, View, f32, vp, vp, vp],
    "cgen_dgauss_sample": [i32, i32, i32, i32, i32, View, f32, vp, u32, vp, vp, vp],
    "cgen_gauss_nll_fwd": [i32, i32, i32, i32, i32, View, View, View, vp, u32, vp, vp],
    "cgen_gauss_nll_bwd": [i32, i32, i32, i32, i32, View, View, View, vp, u32, vp, i32, View, vp],
    "cgen_gauss_sample": [i32, i32, i32, i32, i32, View, f32, vp, u32, vp, vp, vp],
    "cgen_dmol_nll_fwd": [i32, i32, i32, i32, View, View, vp, vp],
    "cgen_dmol_nll_bwd": [i32, i32, i32, i32, View, View, vp, i32, View, vp],
    "cgen_dmol_decode": [i32, i32, i32, i32, View, i32, vp, u32, f32, vp, vp, vp],
    "cgen_elbo_finalize": [i32, vp, i32, f32, vp, i32, f32, f32, vp, vp, vp],
    "cgen_cf_dgauss_bwd": [i32, i32, i32, i32, i32, View, View, View, vp, f32, View, View, vp],
    "cgen_cf_pixels": [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "cgen_sumsq_partial": [vp, i64, vp, i32, vp],
    "cgen_clip_decide": [vp, i32, vp, f32, f32, vp, vp],
    "cgen_adamw_ema": [C.POINTER(AdamwArgs), vp],
    "cgen_step_commit": [vp, vp],
    "cgen_philox_normal": [vp, i64, vp, u32, vp],
    "cgen_rng_advance": [vp, u64, vp],
}
_RESTYPES = {"cgen_last_error": C.c_char_p}
ABI_VERSION = 407  # CGEN_ABI_VERSION of include/cgen_hip.h this binding was written against
_NOCHECK = {"cgen_version", "cgen_h16_format", "cgen_last_error", "cgen_conv2d_wgrad_plan", "cgen_reparam_kl_chunks", "cgen_like_chunks",
            "cgen_block3_supported", "cgen_block4_supported", "cgen_block4_pair_supported", "cgen_block3_pair_supported", "cgen_conv2d_pair_supported", "cgen_stem_conv_supported"}


class WgradBatchLaunch(C.Structure):
    _fields_ = [("ncf", C.c_int32), ("ks", C.c_int32), ("lds_bytes", C.c_int32), ("nblocks", C.c_int32), ("blocks_offset", C.c_int64)]


class CgenError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path):
        self.path = path
        self.cdll = C.CDLL(path)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
            setattr(self, "_raw_" + name, fn)
            setattr(self, name[len("cgen_"):], fn if name in _NOCHECK else self._checked(name, fn))
        if self.cdll.cgen_version() != ABI_VERSION:
            raise CgenError(f"{path}: ABI version {self.cdll.cgen_version()}, this binding needs {ABI_VERSION} -- a stale or foreign build "
                            "(rebuild with causal-gen_amd/build.sh)")
        # 16-bit storage format of THIS build: torch tensors handed over as CGEN_F16 must be in it
        self.h16_is_bf16 = bool(self.cdll.cgen_h16_format())

    def _checked(self, name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise CgenError(f"{name} failed ({rc}): {self.cdll.cgen_last_error().decode()}")
        call.__name__ = name
        return call


_LIB = None


def load():
    """Load libcgen_hip.so (no GPU needed).  Raises with build instructions when it is absent."""
    global _LIB
    if _LIB is None:
        # torch must be in the process first: it bundles its own libamdhip64, and the kernels here must launch
        # through the SAME HIP runtime instance that owns torch's streams and allocations.
        import torch  # noqa: F401

        if not os.path.exists(LIB_PATH):
            raise CgenError(f"{LIB_PATH} not found: build it with causal-gen_amd/build.sh "
                            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _LIB = _Lib(LIB_PATH)
    return _LIB


def require_gpu():
    """The product path runs on an MI355X only; fail loudly otherwise."""
    import torch

    lib = load()
    if not torch.cuda.is_available():
        raise CgenError("causal-gen_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False and there is "
                        "no CPU fallback by design")
    return lib
