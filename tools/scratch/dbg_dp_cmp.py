import sys, torch
a, b = torch.load(f"gpurun_out/dp_{sys.argv[1]}.pt"), torch.load(f"gpurun_out/dp_{sys.argv[2]}.pt")
n = 0
for k in a["sd"]:
    if not torch.equal(a["sd"][k], b["sd"][k]):
        n += 1
        if n < 15:
            print("DIFF", k, float((a["sd"][k] - b["sd"][k]).abs().max()))
print("params differing:", n, "of", len(a["sd"]), "| grads equal:", torch.equal(a["g"], b["g"]), "| early", b["early"])
d = (a["g"] - b["g"]).abs()
nz = d.nonzero().flatten()
if nz.numel():
    print("grad diffs at flat offsets", int(nz[0]), "..", int(nz[-1]), "count", nz.numel())
