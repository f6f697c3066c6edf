#!/usr/bin/env python3
"""Resource / instruction summary of one kernel in tol_amd/lib/kernels.gfx950.s (`make -C tol_amd/csrc asm`).
usage: tools/isa_report.py [mangled-name-substring]   (default: the fp64/S10/shear/reference fg_kernel)"""
import re
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "fg_kernelIdLi0ELi1ELi2ELi0E"
s = open("tol_amd/lib/kernels.gfx950.s").read()
for m in re.finditer(r"^(_ZN5tolfg\S*" + re.escape(pat) + r"\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    body = m.group(2)
    print(m.group(1))
    for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
        mm = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        print("  %-28s %s" % (k, mm and mm.group(1)))
    code = body.split(".section")[0]
    ops = {}
    for line in code.split("\n"):
        t = line.strip().split()
        if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
            ops[t[0]] = ops.get(t[0], 0) + 1
    def count(prefix):
        return sum(v for k, v in ops.items() if k.startswith(prefix))
    print("  instructions %d: v_*f64 %d, v_* %d, s_* %d, ds_* %d, global_load %d, global_store %d, global_atomic %d, scratch %d"
          % (sum(ops.values()), sum(v for k, v in ops.items() if k.startswith("v_") and "f64" in k), count("v_"), count("s_"),
             count("ds_"), count("global_load"), count("global_store"), count("global_atomic"), count("scratch_")))
    for line in code.split("\n"):
        if " sc1" in line or "global_atomic" in line or "buffer_wbl2" in line or "buffer_inv" in line:
            print("    " + line.strip())
