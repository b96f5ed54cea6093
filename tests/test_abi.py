"""The C-ABI library loads without a GPU and exports every symbol include/cgen_hip.h declares; the ctypes binding
covers exactly that set."""
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "cgen_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgen_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from causal_gen_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib.cdll, s), f"{s} declared in cgen_hip.h but not exported by libcgen_hip.so"
    assert sorted(_lib.PROTOTYPES) == syms
    assert lib.version() >= 100
    assert lib.last_error() is not None


def test_struct_layouts_match_header():
    """sizeof() of the ctypes mirrors == the C structs (checked against a tiny C program compiled with gcc)."""
    import ctypes
    import subprocess
    import tempfile

    from causal_gen_amd import _lib

    prog = r'''
#include <stdio.h>
#include "cgen_hip.h"
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(cgen_view), sizeof(cgen_conv_args), sizeof(cgen_wgrad_args),
                         sizeof(cgen_wprep_desc), sizeof(cgen_wred_desc), sizeof(cgen_adamw_args), sizeof(cgen_block3_args), sizeof(cgen_block4_args)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (_lib.View, _lib.ConvArgs, _lib.WgradArgs, _lib.WprepDesc, _lib.WredDesc, _lib.AdamwArgs, _lib.Block3Args, _lib.Block4Args)]
    assert sizes == mine, (sizes, mine)


def test_pure_queries_and_arg_validation_without_gpu():
    from causal_gen_amd import _lib

    lib = _lib.load()
    assert lib.reparam_kl_chunks(96, 96, 16) == 72
    assert lib.like_chunks(192, 192) == 36
    import ctypes

    w = _lib.WgradArgs()
    w.dtype, w.n, w.h, w.w, w.ks, w.nseg = 1, 32, 192, 192, 3, 1
    w.seg[0] = _lib.View(4096, 192 * 192 * 32, 192 * 32, 32, 32, 0)
    w.gout = _lib.View(8192, 192 * 192 * 8, 192 * 8, 8, 8, 0)
    tiled = ctypes.c_int32(-1)
    # 32 -> 8 channels, 3x3: the streaming kernel (round 5) takes it with grad_out (the narrow operand) as the shifted one => layout 1
    assert lib.conv2d_wgrad_plan(ctypes.byref(w), ctypes.byref(tiled)) >= 1 and tiled.value == 3
    w.seg[0], w.gout = _lib.View(8192, 192 * 192 * 8, 192 * 8, 8, 8, 0), _lib.View(4096, 192 * 192 * 32, 192 * 32, 32, 32, 0)
    assert lib.conv2d_wgrad_plan(ctypes.byref(w), ctypes.byref(tiled)) >= 1 and tiled.value == 2  # 8 -> 32: X is the narrow operand
    w.ks = 7  # the 7x7 stem (one input channel, zero padded to 8): X is the shifted operand, 49 taps x 8 = 13 fragments
    w.seg[0] = _lib.View(4096, 192 * 192 * 8, 192 * 8, 8, 1, 8)
    assert lib.conv2d_wgrad_plan(ctypes.byref(w), ctypes.byref(tiled)) >= 1 and tiled.value == 2
    w.dtype = 0  # f32: the generic kernel
    assert lib.conv2d_wgrad_plan(ctypes.byref(w), ctypes.byref(tiled)) >= 1 and tiled.value == 0
    # argument validation happens before any launch, so it is testable on a CPU-only host
    a = _lib.ConvArgs()
    a.dtype = 7
    try:
        lib.conv2d(ctypes_byref(a), None)
        raise AssertionError("expected CgenError")
    except _lib.CgenError as e:
        assert "dtype" in str(e)


def ctypes_byref(x):
    import ctypes

    return ctypes.byref(x)


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from causal_gen_amd import _lib, vae
    from causal_gen_amd.hps import setup_hparams

    m = vae.HVAE(setup_hparams("morphomnist"))
    x = torch.zeros(1, 1, 32, 32)
    pa = torch.zeros(1, 12, 32, 32)
    with pytest.raises(_lib.CgenError):
        m(x, pa)
    with pytest.raises(RuntimeError):
        m.encoder(x)  # holders never compute


def test_device_code_has_no_packed_f32_instructions():
    """MI355X erratum found in round 1 (tools/coexec_probe.py): v_pk_mul/add/fma_f32 in one wave return
    corrupted results while a wave of ANOTHER kernel on the same SIMD executes a 16-bit MFMA (found with v_mfma_f32_16x16x32_bf16; the f16 build keeps the guard) -- the background
    weight-gradient kernel runs next to the whole backward chain.  build.sh switches the packed-fp32 feature off; this
    checks the shipped library instead of the flag."""
    import glob
    import os
    import re
    import shutil
    import subprocess
    import tempfile

    import pytest

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "causal-gen_amd", "libcgen_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("llvm-objdump or the built library is missing")
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(so, d)
        subprocess.run([objdump, "--offloading", os.path.join(d, "libcgen_hip.so")], check=True, capture_output=True)
        bundles = glob.glob(os.path.join(d, "*gfx950*"))
        assert bundles, "no gfx950 code object in the library"
        n_inst = 0
        for b in bundles:
            asm = subprocess.run([objdump, "-d", b], check=True, capture_output=True, text=True).stdout
            n_inst += asm.count("v_mfma_f32_16x16x32_f16") + asm.count("v_mfma_f32_16x16x32_bf16")
            bad = re.findall(r"v_pk_(?:mul|add|fma)_f32", asm)
            assert not bad, "%d packed-f32 instructions in %s" % (len(bad), os.path.basename(b))
        assert n_inst > 0  # we did look at the real kernels


def test_hand_scheduled_kernels_keep_their_scratch_budget():
    """the fused Block kernel (csrc/block.hip) and the streaming weight-gradient kernel (csrc/wgrad3.hip) order their
    untracked LDS-DMA requests by counted `s_waitcnt vmcnt(N)`.  A compiler-made VMEM operation (a scratch access) between them
    cannot break that -- requests retire in order, so an extra counted operation only makes a wait cover MORE of the older
    requests -- but it costs time on the chain, and a spill inside a tile loop would be a performance bug nobody sees.  This pins
    the state of the shipped library: every blk3 instance has its 20-byte private array (5 dwords, indexed dynamically) and
    nothing else, except four instances with 8-14 spilled registers (three remainder-plane ones, one streaming data-gradient
    one); the weight-gradient kernels keep at most a handful of COLD values (entry / epilogue) in scratch.  Growth fails here."""
    import glob
    import os
    import re
    import shutil
    import subprocess
    import tempfile

    import pytest

    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "causal-gen_amd", "libcgen_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(so)):
        pytest.skip("llvm tools or the built library are missing")
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(so, d)
        subprocess.run([objdump, "--offloading", os.path.join(d, "libcgen_hip.so")], check=True, capture_output=True)
        seen = {"blk3": 0, "wg3": 0}
        spilling = []
        for b in glob.glob(os.path.join(d, "*gfx950*")):
            notes = subprocess.run([readelf, "--notes", b], check=True, capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", notes, re.S):
                name, body = m.group(1), m.group(2)
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
                sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", body)
                spills = int(sp.group(1)) if sp else 0
                if "blk3_kernel" in name:
                    seen["blk3"] += 1
                    assert scratch <= 80 and spills <= 16, (name, scratch, spills)
                    if spills:
                        spilling.append(name)
                if "wg3_" in name:
                    seen["wg3"] += 1
                    assert scratch <= 64 and spills <= 8, (name, scratch, spills)
        assert seen["blk3"] >= 10 and seen["wg3"] >= 2, seen
        assert len(spilling) <= 4, spilling
