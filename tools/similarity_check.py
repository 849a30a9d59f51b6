"""Developer aid: how close is any file of this repository to a same-named file of the reference tree?  (The product is a
re-design, not a port; the host-side mirror keeps the reference's class / method / state_dict names on purpose, so this
is worth watching.)  Prints the highest character-, line- and token-level difflib ratios.
usage: python tools/similarity_check.py [reference_root]   (default /root/reference; skipped quietly when absent)"""
import difflib, glob, os, sys

ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not os.path.isdir(ref_root):
    sys.exit(f"{ref_root} not present: nothing to compare against")
EXT = (".py", ".h", ".hip", ".c", ".cpp", ".cu", ".cuh")
by_name = {}
for f in glob.glob(os.path.join(ref_root, "**", "*"), recursive=True):
    if os.path.isfile(f) and f.endswith(EXT):
        by_name.setdefault(os.path.basename(f), []).append(f)


def lines(s):
    return [l.strip() for l in s.splitlines() if l.strip() and not l.strip().startswith(("#", "//"))]


rows = []
for f in glob.glob(os.path.join(repo, "**", "*"), recursive=True):
    if not (os.path.isfile(f) and f.endswith(EXT)) or "/gpurun_out/" in f or "/_obj/" in f or "/_build/" in f:
        continue
    for cand in by_name.get(os.path.basename(f), []):
        a, b = open(f, errors="ignore").read(), open(cand, errors="ignore").read()
        if not a.strip() or not b.strip():
            continue
        la, lb = lines(a), lines(b)
        rows.append((max(difflib.SequenceMatcher(None, a, b).ratio(), difflib.SequenceMatcher(None, la, lb).ratio(),
                         difflib.SequenceMatcher(None, " ".join(la).split(), " ".join(lb).split()).ratio()),
                     os.path.relpath(f, repo), os.path.relpath(cand, ref_root)))
for r, mine, theirs in sorted(rows, reverse=True)[:15]:
    print(f"{r:.2f}  {mine}  <->  {theirs}")
