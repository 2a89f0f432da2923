#!/usr/bin/env python3
"""Static instruction count of a loop of a kernel listing (hipcc -S): all basic blocks from the loop's header to the last
block whose annotation names that header ("in Loop: Header=BBf_H" / "Parent Loop BBf_H").  A proxy for the per-iteration
count while editing a loop body (both arms of wave-uniform branches count).
usage: tools/loop_valu.py file.s <kernel substring> [header number | auto]   (auto: the depth-2 loop with the most VALU)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "auto"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start + 1:end]
class L:
    def __init__(s, i, num, ann): s.i, s.num, s.ann = i, num, ann
    def group(s, k): return {2: s.num, 3: s.ann}[k]
labels = []
for i, l in enumerate(body):
    m = re.match(r"\.LBB(\d+)_(\d+):(.*)", l.strip())
    if not m: continue
    ann, j = m.group(3), i + 1
    while j < len(body) and body[j].strip().startswith(";"):       # the annotation continues on comment lines
        ann += " " + body[j].strip(); j += 1
    labels.append((i, L(i, m.group(2), ann)))
def region(h):
    first = next(i for i, m in labels if m.group(2) == h)
    last = first
    for k, (i, m) in enumerate(labels):
        if i > first and re.search(r"BB\d+_%s\b" % h, m.group(3)):
            last = labels[k + 1][0] if k + 1 < len(labels) else len(body)
    return first, last
def count(a, b):
    c = dict(valu=0, f64=0, trans=0, cndmask=0, vmov=0, lane=0, salu=0, smem=0, lds=0, vmem=0, scratch=0)
    for l in body[a:b]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op: c["f64"] += 1
            if re.match(r"v_(rcp|rsq|sqrt)_f64", op): c["trans"] += 1
            if op.startswith("v_readlane") or op.startswith("v_writelane"): c["lane"] += 1
            if op.startswith("v_cndmask"): c["cndmask"] += 1
            if op.startswith("v_mov"): c["vmov"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"): c["smem"] += 1
        elif op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_nop"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("global_") or op.startswith("buffer_"): c["vmem"] += 1
    return c
if want == "auto":
    heads = [m.group(2) for i, m in labels if "Loop Header: Depth=2" in m.group(3)] or [m.group(2) for i, m in labels if "Loop Header: Depth=1" in m.group(3)]
    best = max(heads, key=lambda h: count(*region(h))["valu"])
    want = best
a, b = region(want)
print("loop header", want, "lines", b - a, count(a, b))
