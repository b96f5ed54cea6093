#!/usr/bin/env python3
"""Refuse profile summaries taken on other code: every profiles/<tag>*.json must carry _source.code_tree_sha == tools/tree_sha.py of
this tree.  usage: python tools/check_profiles.py r04   (exit status 1 and a list when something is stale)"""
import glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tools"))
from tree_sha import tree_sha
now, bad = tree_sha(root), []
for f in sorted(glob.glob(os.path.join(root, "profiles", sys.argv[1] + "*.json"))):
    try:
        sha = (json.load(open(f)).get("_source") or {}).get("code_tree_sha")
    except Exception:
        sha = None
    if sha is not None and sha != now:
        bad.append((os.path.basename(f), sha))
print("code tree", now, "--", "all summaries match" if not bad else "STALE: %s" % bad)
sys.exit(1 if bad else 0)
