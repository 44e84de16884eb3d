#!/usr/bin/env python3
"""Per-function code size, registers and scratch of an AMDGPU assembly file (hipcc --cuda-device-only -S): the kernel-resource-usage remarks
only list kernels, and a kernel's numbers are the maximum over the functions it calls -- this shows which function sets them."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []
for m in re.finditer(r"\.size\s+(_Z\w+), \.Lfunc_end\d+-\1\n(.*?)(?=\n\s*\.text|\n\s*\.section\s+\.rodata|\Z)", txt, re.S):
    name, tail = m.group(1), m.group(2)
    g = lambda k: (re.search(r"; %s:? *=? *(\d+)" % k, tail) or [None, "?"])[1]
    rows.append((name, g("codeLenInByte"), g("NumVgprs"), g("NumAgprs"), g("ScratchSize"), g("TotalNumSgprs")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for n, r in zip(names, rows):
    n = re.sub(r"mpcx::(engine|models)::", "", n)
    n = re.sub(r"\(.*", "", n)
    print("%-70s code %6s B  vgpr %4s  agpr %4s  scratch %5s B/lane  sgpr %s" % (n[:70], *r[1:]))
