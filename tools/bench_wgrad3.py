"""Micro-benchmark of the weight-gradient kernels through the C ABI: one problem per launch (the whole chip to itself), HIP-event timed,
reported as algorithmic HBM rate (X + grad_out read once, 2 bytes per element) and MFMA rate of the useful FLOPs.
usage: python tools/bench_wgrad3.py [shape-set]      env: CGEN_WG3=0 -> the round-2..4 tiled kernel for comparison"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from causal_gen_amd import _lib  # noqa: E402

UKBB = [  # (N, H, W, segs, Co, ks, act): the Block shapes of ukbb192 at the bench batch
    (32, 192, 192, [32], 8, 3, 1), (32, 192, 192, [8], 32, 3, 1), (32, 192, 192, [8], 64, 3, 1),
    (32, 96, 96, [64], 16, 3, 1), (32, 96, 96, [16], 64, 3, 1), (32, 96, 96, [16], 96, 3, 1), (32, 96, 96, [64, 8, 64], 16, 3, 1),
    (32, 48, 48, [96], 24, 3, 1), (32, 48, 48, [24], 96, 3, 1), (32, 48, 48, [24], 128, 3, 1), (32, 48, 48, [96, 8, 96], 24, 3, 1),
    (32, 24, 24, [128], 32, 3, 1), (32, 24, 24, [32], 128, 3, 1), (32, 24, 24, [32], 160, 3, 1), (32, 24, 24, [128, 8, 128], 32, 3, 1),
    (32, 12, 12, [160], 40, 3, 1), (32, 12, 12, [40], 160, 3, 1),
    (32, 48, 48, [16, 8], 96, 1, 0), (32, 24, 24, [144], 128, 1, 0),
]


def view(t, c):
    return _lib.View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), c, t.shape[3] if t.shape[3] != c else 0)


def main():
    lib = _lib.require_gpu()
    st = torch.cuda.current_stream().cuda_stream
    tot_t, tot_b = 0.0, 0.0
    for case in UKBB:
        N, H, W, segc, Co, ks, act = case
        xt = [torch.randn(N, H, W, (c + 7) // 8 * 8, device="cuda").half() for c in segc]
        gt = torch.randn(N, H, W, (Co + 7) // 8 * 8, device="cuda").half()
        a = _lib.WgradArgs()
        a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act = 1, N, H, W, ks, len(segc), act
        for k, (t, c) in enumerate(zip(xt, segc)):
            a.seg[k] = view(t, c)
        a.gout = view(gt, Co)
        kind = C.c_int32(-1)
        nsplit = lib.conv2d_wgrad_plan(C.byref(a), C.byref(kind))
        ci = sum(segc)
        nw = Co * ks * ks * ci
        part = torch.empty(nsplit * (nw + Co), dtype=torch.float32, device="cuda")
        a.nsplit, a.partial_w, a.partial_b = nsplit, part.data_ptr(), part.data_ptr() + 4 * nsplit * nw
        for _ in range(3):
            lib.conv2d_wgrad(C.byref(a), st)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.conv2d_wgrad(C.byref(a), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        byts = 2.0 * N * H * W * (ci + Co)
        flop = 2.0 * N * H * W * ci * Co * ks * ks
        tot_t += us
        tot_b += byts
        print("%-34s kind %d nsplit %3d | %8.1f us | %6.2f TB/s algorithmic | %6.1f TF/s | partial %.1f MB" % (
            "%dx%dx%d %s->%d k%d" % (N, H, W, "+".join(map(str, segc)), Co, ks), kind.value, nsplit, us, byts / us / 1e6, flop / us / 1e6,
            4.0 * nsplit * nw / 1e6), flush=True)
    print("sum %.1f us, %.2f TB/s over the set" % (tot_t, tot_b / tot_t / 1e6))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "batch"):
    main()


def _blocks_of(lib, a0):
    """workgroups the packed launch gives one copy of the problem"""
    arr = (_lib.WgradArgs * 1)(a0)
    a = arr[0]
    a.nsplit, a.partial_w, a.partial_b = lib.conv2d_wgrad_plan(C.byref(a0), None), 0x1000, 0x2000
    nbytes, nl = C.c_int64(0), C.c_int32(0)
    elig = (C.c_int32 * 1)()
    lib.conv2d_wgrad_batch_plan(arr, 1, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
    host = (C.c_char * max(nbytes.value, 1))()
    launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
    lib.conv2d_wgrad_batch_plan(arr, 1, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
    return max(1, sum(launches[i].nblocks for i in range(nl.value)))


def batch_main():
    """Full-chip form: R independent copies of ONE shape in one packed launch (>= ~1500 workgroups), so that every CU holds its two
    workgroups for the whole measurement -- the regime of the engine's flush."""
    lib = _lib.require_gpu()
    st = torch.cuda.current_stream().cuda_stream
    only = sys.argv[2:] and [int(v) for v in sys.argv[2].split(",")]
    for idx, case in enumerate(UKBB):
        if only and idx not in only:
            continue
        N, H, W, segc, Co, ks, act = case
        xt = [torch.randn(N, H, W, (c + 7) // 8 * 8, device="cuda").half() for c in segc]
        gt = torch.randn(N, H, W, (Co + 7) // 8 * 8, device="cuda").half()
        a0 = _lib.WgradArgs()
        a0.dtype, a0.n, a0.h, a0.w, a0.ks, a0.nseg, a0.act = 1, N, H, W, ks, len(segc), act
        for k, (t, c) in enumerate(zip(xt, segc)):
            a0.seg[k] = view(t, c)
        a0.gout = view(gt, Co)
        nsplit = lib.conv2d_wgrad_plan(C.byref(a0), None)
        ci = sum(segc)
        nw = Co * ks * ks * ci
        byts = 2.0 * N * H * W * (ci + Co)
        R = max(1, min(64, int(8e9 / byts / 8)))  # ~1 GB of input per launch
        kind0 = C.c_int32(-1)
        lib.conv2d_wgrad_plan(C.byref(a0), C.byref(kind0))
        if kind0.value >= 2 and os.environ.get("BENCH_FILL", "1") == "1":
            # ... rounded so that the launch is a whole number of residency rounds (2 workgroups x 256 CUs): with 640 workgroups the
            # second round runs a quarter full and the rate reads 1.6x low
            bpp = _blocks_of(lib, a0)
            best = None
            for r in range(max(1, R // 2), 2 * R + 1):
                nb = r * bpp
                fill = nb / (512.0 * ((nb + 511) // 512))
                if best is None or fill > best[0] + 1e-9:
                    best = (fill, r)
            R = best[1]
        parts, args = [], []
        for r in range(R):
            part = torch.empty(nsplit * (nw + Co), dtype=torch.float32, device="cuda")
            a = _lib.WgradArgs.from_buffer_copy(bytes(a0))
            a.nsplit, a.partial_w, a.partial_b = nsplit, part.data_ptr(), part.data_ptr() + 4 * nsplit * nw
            parts.append(part)
            args.append(a)
        arr = (_lib.WgradArgs * R)(*args)
        nbytes, nl = C.c_int64(0), C.c_int32(0)
        elig = (C.c_int32 * R)()
        lib.conv2d_wgrad_batch_plan(arr, R, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
        host = (C.c_char * max(nbytes.value, 1))()
        launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
        lib.conv2d_wgrad_batch_plan(arr, R, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
        blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
        nblk = sum(launches[i].nblocks for i in range(nl.value))
        for _ in range(2):
            lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, 0, st)
        torch.cuda.synchronize()
        reps = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, 0, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        flop = 2.0 * N * H * W * ci * Co * ks * ks * R
        print("%-34s x%2d  %5d blocks | %8.1f us | %5.2f TB/s algorithmic | %6.1f TF/s" % (
            "%dx%dx%d %s->%d k%d" % (N, H, W, "+".join(map(str, segc)), Co, ks), R, nblk, us, byts * R / us / 1e6, flop / us / 1e6), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "batch":
    batch_main()
