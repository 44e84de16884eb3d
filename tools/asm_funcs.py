"""Per-function register and scratch use of a device assembly file (hipcc -S --cuda-device-only): the AMDGPU backend's
`.set <symbol>.num_vgpr / .private_seg_size` lines, demangled.  Usage: python tools/asm_funcs.py file.s [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
vg = dict(re.findall(r"\.set \.?L?(_Z\w+)\.num_vgpr, (\d+)", txt))
sc = dict(re.findall(r"\.set \.?L?(_Z\w+)\.private_seg_size, (\d+)", txt))
sz = {m.group(1): int(m.group(2)) for m in re.finditer(r"^(_Z\w+):.*?; codeLenInByte = (\d+)", txt, flags=re.S | re.M)} if False else {}
names = sorted(vg)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    d = re.sub(r"mpcx::(engine|models)::", "", d)
    if want and not all(w in d for w in want):
        continue
    print("%4s vgpr %5s B scratch  %s" % (vg[n], sc.get(n, "?"), d[:150]))
