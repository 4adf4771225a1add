"""Register / occupancy table of every gemm_mfma_kernel instantiation in libnmfx (development aid).
usage: python scripts/kernel_regs.py [extra hipcc flags, e.g. -I some/older/headers]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + sys.argv[1:] + [
    os.path.join(ROOT, "nmf.jl_amd/csrc/nmfx_api.hip"), "-o", "/tmp/_regs.so", "-L/opt/rocm/lib", "-lrccl",
    "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"n": m.group(1)}; rows.append(cur); continue
    for k in ("VGPRs", "AGPRs", "Occupancy [waves/SIMD]", "VGPRs Spill"):
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
        if m and cur is not None: cur[k] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["n"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, names):
    if "gemm_mfma" not in d: continue
    d = d.replace("void nmfx::gemm_mfma_kernel", "").split("(")[0]
    print("%-84s v=%s a=%s occ=%s spill=%s" % (d, r.get("VGPRs"), r.get("AGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill")))
