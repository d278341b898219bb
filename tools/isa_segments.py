#!/usr/bin/env python3
"""usage: isa_segments.py file.hip kernel-substring : per s_barrier segment counts of MFMA / LDS / global / scratch instructions"""
import subprocess, sys, os
src, pat = sys.argv[1], sys.argv[2]
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "discrete_dppm_graphs_amd", "csrc")
out = "/tmp/_seg.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I../../include", "-I.", "-S", "--cuda-device-only", src, "-o", out] + sys.argv[3:], cwd=here, stderr=subprocess.DEVNULL)
s = open(out).read()
names = [l.split(":")[0] for l in s.split("\n") if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
for nm in names:
    i = s.index(nm + ":")
    j = s.index(".end_amdhsa_kernel", i) if ".end_amdhsa_kernel" in s[i:] else len(s)
    j2 = s.find("s_endpgm", i)
    body = s[i:j].split("\n")
    print(nm)
    cur = dict(mfma=0, sst=0, sld=0, dsr=0, dsw=0, gl=0, gs=0, n=0); k = 0
    for l in body:
        t = l.strip(); cur["n"] += 1
        if t.startswith("v_mfma"): cur["mfma"] += 1
        elif t.startswith("scratch_store"): cur["sst"] += 1
        elif t.startswith("scratch_load"): cur["sld"] += 1
        elif t.startswith("ds_read"): cur["dsr"] += 1
        elif t.startswith("ds_write"): cur["dsw"] += 1
        elif t.startswith("global_load"): cur["gl"] += 1
        elif t.startswith("global_store"): cur["gs"] += 1
        elif t.startswith("s_barrier"):
            print(" ", k, cur); k += 1
            cur = dict(mfma=0, sst=0, sld=0, dsr=0, dsw=0, gl=0, gs=0, n=0)
    print(" ", k, cur)
