#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as the
MI355X guide prescribes).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is taken as reported.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import collections
import csv
import json
import sys

CLASSES = {"conv_fwd+dgrad": ("conv_tile_kernel", "conv_ws_kernel", "conv_px_kernel", "conv_smallp_kernel", "conv_smallp_pair_kernel", "conv_kernel", "blk3", "blk4_kernel"), "conv_wgrad": ("wgrad_tile_kernel", "wgrad_tile_batched_kernel", "wgrad_tile_mega_kernel", "wgrad_kernel", "wg3_mega_kernel", "wg3_single_kernel"),
           "wgrad_reduce": ("wred_kernel",), "reparam_kl": ("reparam_kl",), "elementwise": ("axpby", "avgpool", "upsample", "im2col", "batch_")}


def load(path, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        cls = next((c for c, keys in CLASSES.items() if any(k in name for k in keys)), "other")
        per[cls][0] += float(r["Counter_Value"])
        per[cls][1] += 1
    return per


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for cls in sorted(set(f) | set(w)):
        fk, n = f.get(cls, [0.0, 0])
        wk, _ = w.get(cls, [0.0, 0])
        out[cls] = dict(dispatches=n, fetch_kb_raw=fk, write_kb_raw=wk, hbm_bytes_total=(2 * fk + wk) * 1024,
                        hbm_bytes_per_dispatch=((2 * fk + wk) * 1024 / n) if n else None)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tree_sha import tree_sha
    out["_source"] = {"code_tree_sha": tree_sha(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --prep-steps 1 --no-cpu --no-cf --no-f32 --no-extra"}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
