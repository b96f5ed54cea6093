"""Register / scratch / occupancy table of the kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_regs.py causal-gen_amd/csrc/conv.hip [name-substring]"""
import re, subprocess, sys, os
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", os.path.basename(src), "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], cwd=os.path.dirname(os.path.abspath(src)), capture_output=True, text=True)
cur, rows = None, []
keys = {"VGPRs": "v", "AGPRs": "a", "SGPRs": "s", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ"}
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"n": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(r"remark:\s*" + re.escape(k) + r": (\d+)", l)
        if m and cur is not None:
            cur[short] = int(m.group(1))
for r_ in rows:
    if pat in r_["n"]:
        name = subprocess.run(["c++filt", r_["n"]], capture_output=True, text=True).stdout.strip()
        print("%-90s v %3d a %3d s %3d scratch %4d occ %d" % (name[:90], r_.get("v", -1), r_.get("a", -1), r_.get("s", -1), r_.get("scratch", -1), r_.get("occ", -1)))
