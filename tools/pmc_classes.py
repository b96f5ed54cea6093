#!/usr/bin/env python3
"""Per kernel CLASS summary of the two instruction-mix PMC passes (tools/pmc_insts.sh): MFMA-pipe busy share and instructions per
MFMA.  mfma_busy_pct_of_wave_lifetime = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) (the guide: MFMA_BUSY counts cycles,
WAVE_CYCLES quad-cycles): the share of a wave's resident time its SIMD's matrix pipe was busy -- an UPPER bound on the pipe's
utilisation when several waves share a SIMD.   usage: pmc_classes.py gpurun_out/<tag>"""
import collections
import csv
import json
import sys

CLASSES = {"conv_fwd+dgrad": ("conv_tile_kernel", "conv_ws_kernel", "conv_px_kernel", "conv_smallp_kernel", "conv_smallp_pair_kernel", "conv_kernel", "stem7", "blk3", "blk4_kernel"),
           "conv_wgrad": ("wgrad_tile", "wgrad_kernel", "wg3_"), "wgrad_reduce": ("wred_kernel",), "reparam_kl": ("reparam_kl",),
           "likelihood": ("dgauss", "dmol"), "optimizer": ("adamw", "sumsq", "clip_decide")}
tag = sys.argv[1]
base = tag.split("/")[-1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for sfx in ("A", "B"):
    for r in csv.DictReader(open(f"{tag}_{sfx}/{base}_{sfx}_counter_collection.csv")):
        name = r["Kernel_Name"]
        cls = next((c for c, keys in CLASSES.items() if any(k in name for k in keys)), "other")
        tot[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_MFMA"):
            tot[cls]["dispatches_" + sfx] += 1
out = {}
for cls, a in sorted(tot.items()):
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    mf = a.get("SQ_INSTS_MFMA", 0.0)
    out[cls] = {"dispatches": int(a.get("dispatches_A", 0)),
                "mfma_busy_pct_of_wave_lifetime": (100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc)) if wc else None,
                "valu_per_mfma": a.get("SQ_INSTS_VALU", 0.0) / mf if mf else None, "salu_per_mfma": a.get("SQ_INSTS_SALU", 0.0) / mf if mf else None,
                "lds_per_mfma": a.get("SQ_INSTS_LDS", 0.0) / mf if mf else None,
                "issue_stall_pct_of_wave_lifetime": (100.0 * a.get("SQ_WAIT_INST_ANY", 0.0) / wc) if wc else None}
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tree_sha import tree_sha
out["_source"] = {"code_tree_sha": tree_sha(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "command": "tools/pmc_insts.sh (two SQ counter passes of one bench step)"}
print(json.dumps(out, indent=1))
