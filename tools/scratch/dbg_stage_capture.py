import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from causal_gen_amd.train import TrainStep
name = sys.argv[1] if len(sys.argv) > 1 else "ukbb192"
m, hp = bench.build_model(name, "f16")
m = m.cuda()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ts = TrainStep(m, hp, ema=True, use_graph=True)
x, pa = bench.synth_batch(name, hp, B, torch.device("cuda"), 100)
eng = m.engine()
import causal_gen_amd.stage as st
orig = st.StageMixin.stage_flush
log = []
def patched(self):
    if self._stage_ops:
        import ctypes as C
        key = b"".join(bytes(C.c_int32(k)) + bytes(a) for k, a in self._stage_ops)
        log.append((hash(key), len(self._stage_ops), torch.cuda.is_current_stream_capturing(), [(k, bytes(a)) for k, a in self._stage_ops]))
    return orig(self)
st.StageMixin.stage_flush = patched
type(eng).stage_flush = patched
try:
    ts.step(x, pa)
except Exception as e:
    print("EXC", e)
eager = [l for l in log if not l[2]]
cap = [l for l in log if l[2]]
print(len(eager), len(cap))
for i, (a, b) in enumerate(zip(eager, cap)):
    if a[0] != b[0]:
        print("first differing list", i, a[1], b[1])
        for j, ((k1, b1), (k2, b2)) in enumerate(zip(a[3], b[3])):
            if b1 != b2:
                d = [t for t in range(min(len(b1), len(b2))) if b1[t] != b2[t]]
                print(" op", j, "kind", k1, k2, "differing bytes at", d[:24], "len", len(b1))
                break
        break
