#!/usr/bin/env python3
"""Per-kernel instruction mix from the two passes of tools/pmc_insts.sh: per-wave VALU / SALU / LDS / VMEM / MFMA counts.
usage: pmc_insts.py gpurun_out/<tag>  [substring filter]"""
import collections
import csv
import sys

tag = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sfx in ("A", "B"):
    base = tag.split("/")[-1]
    for r in csv.DictReader(open(f"{tag}_{sfx}/{base}_{sfx}_counter_collection.csv")):
        if filt in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void cgen::", "").split("(")[0]
            agg[(name, r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    w = a.get("SQ_WAVES", 1) or 1
    print("%-44s grid %-8s lds %-6s vgpr %-4s waves %6d | per wave: VALU %6.0f SALU %6.0f LDS %5.0f VMEM_RD %5.1f VMEM_WR %5.1f MFMA %5.0f | wave_cycles/wave %7.0f" % (
        k[0][:44], k[1], k[2], k[3], w, a.get("SQ_INSTS_VALU", 0) / w, a.get("SQ_INSTS_SALU", 0) / w, a.get("SQ_INSTS_LDS", 0) / w,
        a.get("SQ_INSTS_VMEM_RD", 0) / w, a.get("SQ_INSTS_VMEM_WR", 0) / w, a.get("SQ_INSTS_MFMA", 0) / w, a.get("SQ_WAVE_CYCLES", 0) / w))
