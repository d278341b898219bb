#!/usr/bin/env python3
"""Scan a built libdppm_hip.so: per kernel, packed-f32 instructions with an op_sel bit set (the form that misbehaves),
with op_sel_hi only, and scratch bytes per lane.  usage: tools/isa_scan.py [lib.so]"""
import subprocess, re, sys, tempfile, pathlib, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import test_isa_gate as g
if len(sys.argv) > 1: g.LIB = sys.argv[1]
tmp = pathlib.Path(tempfile.mkdtemp())
cos = g._code_objects(tmp)
tot_hi = tot_lo = 0
for co in cos:
    dis = subprocess.check_output([f"{g.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", str(co)], text=True)
    for m in re.finditer(r"<(\S+)>:\n(.*?)(?=\n\n|\Z)", dis, re.S):
        hi = re.findall(r"v_pk_(?:mul|fma|add)_f32[^\n]*op_sel:\[[0-9,]*1", m.group(2))
        lo = [x for x in re.findall(r"v_pk_(?:mul|fma|add)_f32[^\n]*op_sel_hi[^\n]*", m.group(2)) if not re.search(r"op_sel:\[[0-9,]*1", x)]
        tot_hi += len(hi); tot_lo += len(lo)
        if hi: print("%4d op_sel-bit  %4d op_sel_hi-only  %s" % (len(hi), len(lo), m.group(1)[:100]))
print("total: %d packed-f32 instructions with an op_sel bit set, %d with op_sel_hi only" % (tot_hi, tot_lo))
