#!/usr/bin/env python3
"""Content hash of the code a profile was taken on: every source file of the package (causal-gen_amd/*.py, csrc/*) and the C
header, in name order.  The GPU box has no .git, so this -- not a commit id -- is what stamps the summaries under profiles/
(`_source.code_tree_sha`) and what bench.py compares them with at run time (`stale`)."""
import glob
import hashlib
import os
import sys


def tree_sha(root):
    files = sorted(glob.glob(os.path.join(root, "causal-gen_amd", "*.py")) + glob.glob(os.path.join(root, "causal-gen_amd", "csrc", "*"))
                   + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha1()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(tree_sha(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
