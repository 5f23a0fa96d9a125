#!/usr/bin/env python3
"""Compact view of a kernel's instruction stream from hipcc -S output: one letter per instruction
(M mfma, T transcendental, V other VALU, D ds_read, W ds_write, G global/buffer load, g global store, w s_waitcnt,
B s_barrier, b branch, s other scalar), broken at labels.  Usage: isa_shape.py file.s kernel-name-substring"""
import re
import sys


def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "T"
    if op.startswith("v_"): return "V"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "D"
    if op.startswith("ds_"): return "W"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): return "G"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")): return "g"
    if op.startswith("s_waitcnt"): return "w"
    if op.startswith("s_barrier"): return "B"
    if op.startswith(("s_cbranch", "s_branch")): return "b"
    if op.startswith("s_"): return "s"
    return "?"


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().endswith(":") is False and ":" in l)
    out, cur, label = [], [], "entry"
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        if re.match(r"^\.LBB\d+_\d+:", s):
            out.append((label, "".join(cur)))
            cur, label = [], s.split(":")[0]
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = cls(op)
        if c == "w":
            m = re.search(r"vmcnt\((\d+)\)", s)
            n = re.search(r"lgkmcnt\((\d+)\)", s)
            c = "w" + ("v%s" % m.group(1) if m else "") + ("l%s" % n.group(1) if n else "") + " "
        cur.append(c)
    out.append((label, "".join(cur)))
    for lab, seq in out:
        cnt = {k: seq.count(k) for k in "MTVDWGgBb"}
        print("%-10s n=%-4d %s" % (lab, len(seq), " ".join("%s%d" % kv for kv in cnt.items() if kv[1])))
        if "--full" in sys.argv:
            print("   ", seq)


main()
