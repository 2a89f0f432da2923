#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one flavour.
usage: tools/kernel_resources.py fast|faithful [filter] [-- extra hipcc flags]"""
import re, subprocess, sys, os
fl = sys.argv[1] if len(sys.argv) > 1 else "fast"
flt = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "--" else ""
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
src = os.path.join(os.path.dirname(__file__), "..", "unconfined_amd", "csrc", f"ucf_kernels_{fl}.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-ffp-contract={'fast' if fl == 'fast' else 'off'}",
       *extra, "-c", src, "-o", f"/tmp/kres_{fl}.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", ln)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
def short(n):
    m = re.match(r"_ZN\d+ucf_\w+?\d+([a-z_]+?)(ILi(\d)ELi(\d)EE|ILi(\d)EE)?E?v?14ucf|_ZN\d+ucf_\w+?\d+([a-z_]+)", n)
    m2 = re.search(r"(\w+_kernel)(ILi(\d)(ELi(\d))?E)?", n)
    if m2: return m2.group(1).split("ucf_")[-1].lstrip("0123456789") + (f"<{m2.group(3)}{',' + m2.group(5) if m2.group(5) else ''}>" if m2.group(3) else "")
    return n[:40]
print(f"{'kernel':34s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'SGPRsp':>6s} {'occ':>3s}")
for r in rows:
    nm = short(r["name"])
    if flt and flt not in nm: continue
    print(f"{nm:34s} {r.get('VGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('SGPRs Spill','?'):>6s} {r.get('Occupancy [waves/SIMD]','?'):>3s}")
