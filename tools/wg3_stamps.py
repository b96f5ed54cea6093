"""Cycle stamps of the streaming weight-gradient kernel (csrc/wgrad3.hip, CGEN_WG3_STAMPS): per tile of the launch's first workgroup,
for each of its 4 waves: [wait for the tile's DMA | barrier | issue of the tile two ahead | K-steps (fragment reads + MFMAs)].
usage: python tools/wg3_stamps.py N H W segs Co ks act      e.g. 32 48 48 96 24 3 1"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
buf = torch.zeros(256, dtype=torch.int64, device="cuda")
os.environ["CGEN_WG3_STAMPS"] = hex(buf.data_ptr())
from causal_gen_amd import _lib  # noqa: E402
from bench_wgrad3 import view  # noqa: E402


def main():
    N, H, W = (int(v) for v in sys.argv[1:4])
    segc = [int(v) for v in sys.argv[4].split("+")]
    Co, ks, act = (int(v) for v in sys.argv[5:8])
    lib = _lib.require_gpu()
    st = torch.cuda.current_stream().cuda_stream
    xt = [torch.randn(N, H, W, (c + 7) // 8 * 8, device="cuda").half() for c in segc]
    gt = torch.randn(N, H, W, (Co + 7) // 8 * 8, device="cuda").half()
    a = _lib.WgradArgs()
    a.dtype, a.n, a.h, a.w, a.ks, a.nseg, a.act = 1, N, H, W, ks, len(segc), act
    for k, (t, c) in enumerate(zip(xt, segc)):
        a.seg[k] = view(t, c)
    a.gout = view(gt, Co)
    nsplit = lib.conv2d_wgrad_plan(C.byref(a), None)
    nw = Co * ks * ks * sum(segc)
    part = torch.empty(nsplit * (nw + Co), dtype=torch.float32, device="cuda")
    a.nsplit, a.partial_w, a.partial_b = nsplit, part.data_ptr(), part.data_ptr() + 4 * nsplit * nw
    R = int(os.environ.get("STAMP_COPIES", "0"))
    if R:  # full-chip regime: R copies of the problem in one packed launch; the stamps are those of the launch's first workgroup
        keep, args = [], []
        for r in range(R):
            pr = torch.empty(nsplit * (nw + Co), dtype=torch.float32, device="cuda")
            ar = _lib.WgradArgs.from_buffer_copy(bytes(a))
            ar.partial_w, ar.partial_b = pr.data_ptr(), pr.data_ptr() + 4 * nsplit * nw
            keep.append(pr)
            args.append(ar)
        arr = (_lib.WgradArgs * R)(*args)
        nbytes, nl = C.c_int64(0), C.c_int32(0)
        elig = (C.c_int32 * R)()
        lib.conv2d_wgrad_batch_plan(arr, R, None, 0, C.byref(nbytes), None, 0, C.byref(nl), elig)
        host = (C.c_char * max(nbytes.value, 1))()
        launches = (_lib.WgradBatchLaunch * max(nl.value, 1))()
        lib.conv2d_wgrad_batch_plan(arr, R, host, nbytes.value, C.byref(nbytes), launches, nl.value, C.byref(nl), elig)
        blob = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
        for _ in range(3):
            lib.conv2d_wgrad_batch_run(blob.data_ptr(), launches, nl.value, 0, st)
    else:
        for _ in range(3):
            lib.conv2d_wgrad(C.byref(a), st)
    torch.cuda.synchronize()
    s = buf.cpu().view(4, 64).tolist()
    t0 = min(w[0] for w in s)
    for wv in range(4):
        w = s[wv]
        print("wave %d: setup %d cycles" % (wv, w[1] - w[0]))
        k = 2
        prev = w[1]
        tile = 0
        while k + 3 < 62 and w[k + 3]:
            print("   tile %2d: wait %5d | barrier %5d | issue %5d | compute %5d | total %5d" % (
                tile, w[k] - prev, w[k + 1] - w[k], w[k + 2] - w[k + 1], w[k + 3] - w[k + 2], w[k + 3] - prev))
            prev = w[k + 3]
            k += 4
            tile += 1


if __name__ == "__main__":
    main()
