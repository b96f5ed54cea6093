"""Per-phase wall-clock stamps (100 MHz) of workgroup 0 for every op of ONE stage launch: CGEN_STAGE_STAMPS.
usage: python tools/stage_stamps.py <config> <batch> [which launch]"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "morphomnist"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m, hp = bench.build_model(name, "f16")
m = m.cuda().eval()
x, pa = bench.synth_batch(name, hp, B, torch.device("cuda"), 100)
buf = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
with torch.no_grad():
    m(x, pa, beta=1.0)
    m(x, pa, beta=1.0)
torch.cuda.synchronize()
eng = m.engine()
import causal_gen_amd.stage as st
count = [0]
orig = st.StageMixin.stage_flush
def patched(self):
    if self._stage_ops:
        count[0] += 1
        if count[0] == which:
            os.environ["CGEN_STAGE_STAMPS"] = str(buf.data_ptr())
            ops = [(o[0], o[1]) for o in self._stage_ops]
            r = orig(self)
            torch.cuda.synchronize()
            os.environ.pop("CGEN_STAGE_STAMPS")
            t = buf.cpu().view(256, 8)
            names = ["start", "dma issued", "landed", "act done", "mfma done", "epi done", "barrier"]
            tot0 = None
            for i, (k, a) in enumerate(ops[:256]):
                row = t[i].tolist()
                if tot0 is None:
                    tot0 = row[0]
                if k == 0:
                    d = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4], row[6] - row[5]]
                    print("op %3d conv%d [%s->%d %dx%d act%d] t=%7.2f us | dma-issue %5.2f wait %5.2f act %5.2f mfma %5.2f epi %5.2f barrier %5.2f | total %5.2f" % (
                        i, a.ks, "+".join(str(a.seg[j].c) for j in range(a.nseg)), a.out.c, a.h, a.w, a.act, (row[0] - tot0) / 100.0,
                        *[v / 100.0 for v in d], (row[6] - row[0]) / 100.0))
                else:
                    print("op %3d kind %d [%dx%d] t=%7.2f us | total %5.2f" % (i, k, a.h, a.w, (row[0] - tot0) / 100.0, (row[6] - row[0]) / 100.0))
            return r
    return orig(self)
type(eng).stage_flush = patched
with torch.no_grad():
    m(x, pa, beta=1.0)
torch.cuda.synchronize()
