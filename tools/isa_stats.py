#!/usr/bin/env python3
"""Instruction mix of one kernel of an ISA listing (hipcc -S --cuda-device-only): totals and per basic block.
usage: python tools/isa_stats.py listing.s 'demangled substring' [--blocks]"""
import re
import subprocess
import sys

path, want = sys.argv[1], sys.argv[2]
show_blocks = "--blocks" in sys.argv
text = open(path).read()
names = sorted(set(re.findall(r"^(_Z\w+):", text, re.M)))
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
hits = [n for n, d in zip(names, dem) if want in d]
if len(hits) != 1:
    print("matches:", [d for d in dem if want in d][:20])
    sys.exit(1)
name = hits[0]
start = text.index("\n" + name + ":")
end = text.index(".Lfunc_end", start)
body = text[start:end].splitlines()


def kind(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, cur, label = [], {}, "entry"
tot = {}
for line in body:
    line = line.strip()
    m = re.match(r"^(\.LBB\w+):", line)
    if m:
        blocks.append((label, cur))
        cur, label = {}, m.group(1)
        continue
    if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
        continue
    op = line.split()[0]
    k = kind(op)
    cur[k] = cur.get(k, 0) + 1
    tot[k] = tot.get(k, 0) + 1
blocks.append((label, cur))
print(name)
print("total:", dict(sorted(tot.items())))
if show_blocks:
    for lab, c in blocks:
        n = sum(c.values())
        if n >= 40:
            print(f"  {lab:12s} {n:5d}  ", dict(sorted(c.items())))
