#!/usr/bin/env python3
"""Resource / instruction summary of one kernel in tol_amd/lib/kernels.gfx950.s (`make -C tol_amd/csrc asm`).
usage: tools/isa_report.py [mangled-name-substring] [asm file]
   default: the fp64 / mixed / shear / reference / non-temporal fg_kernel of the headline, fg_kernelIdLi2ELi1ELi2ELi0ELb1ELi1E
   (template arguments: element type d|f, mission 0 S10 | 1 G7 | 2 mixed, wind 0..3, vector width, pattern 0|1, nt, nodes per lane);
   the packed fp32 kernels live in their own translation unit:
     hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTOLFG_TU=2 -S --cuda-device-only -o /tmp/f32p.s tol_amd/csrc/kernels.hip
     tools/isa_report.py fg_kernelIfLi2ELi1ELi4ELi0ELb1ELi2E /tmp/f32p.s"""
import re
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "fg_kernelIdLi2ELi1ELi2ELi0ELb1ELi1E"
s = open(sys.argv[2] if len(sys.argv) > 2 else "tol_amd/lib/kernels.gfx950.s").read()
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
    print("  packed fp32: v_pk_fma_f32 %d, v_pk_mul_f32 %d, v_pk_add_f32 %d, other v_pk_* %d"
          % (ops.get("v_pk_fma_f32", 0), ops.get("v_pk_mul_f32", 0), ops.get("v_pk_add_f32", 0),
             count("v_pk_") - ops.get("v_pk_fma_f32", 0) - ops.get("v_pk_mul_f32", 0) - ops.get("v_pk_add_f32", 0)))
    print("  instructions %d: v_*f64 %d, v_* %d, s_* %d, ds_* %d, global_load %d, global_store %d, global_atomic %d, scratch %d"
          % (sum(ops.values()), sum(v for k, v in ops.items() if k.startswith("v_") and "f64" in k), count("v_"), count("s_"),
             count("ds_"), count("global_load"), count("global_store"), count("global_atomic"), count("scratch_")))
    for line in code.split("\n"):
        if " sc1" in line or "global_atomic" in line or "buffer_wbl2" in line or "buffer_inv" in line:
            print("    " + line.strip())
