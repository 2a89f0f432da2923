#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing: per basic block, and summed over the body.
usage: tools/isa_mix.py file.s <substring of the kernel's mangled name> [--blocks]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cat(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"): return "fma64"
    if op.startswith("v_mul_f64"): return "mul64"
    if op.startswith("v_add_f64"): return "add64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f64", op): return "trans64"
    if re.match(r"v_(ldexp|frexp|rndne|trunc|floor|fract|cvt|max|min|cmp|cmpx|div)\w*f64", op) or ("f64" in op): return "other64"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "vmov"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "lane"
    if op.startswith("v_cmp"): return "vcmp"
    if op.startswith("v_"): return "valu_other"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "misc"
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"\.LBB\d+_\d+:", t):
            blocks.append((name, cur)); cur, name = collections.Counter(), t.split(":")[0]
        continue
    if re.match(r"\.?[A-Za-z_0-9$]+:", t):
        blocks.append((name, cur)); cur, name = collections.Counter(), t.split(":")[0]
        continue
    cur[cat(t.split()[0])] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks: tot.update(c)
valu = ("fma64", "mul64", "add64", "trans64", "other64", "cndmask", "vmov", "lane", "vcmp", "valu_other")
print("kernel", lines[start][:90], "lines", end - start)
print("TOTAL", dict(tot), "VALU", sum(tot[k] for k in valu))
if "--blocks" in sys.argv:
    for n, c in blocks:
        v = sum(c[k] for k in valu)
        if v >= 20: print(f"{n:12s} VALU {v:4d}", dict(c))
