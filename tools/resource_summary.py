"""One line per kernel from `hipcc -Rpass-analysis=kernel-resource-usage` remarks on stdin (make -C libmpc_amd/csrc resource / resource-nlmpc 2>&1 |
python tools/resource_summary.py): registers, spills, scratch, occupancy -- the form of profiles/rNN_resource_usage_*.txt."""
import re
import subprocess
import sys

txt = sys.stdin.read()
for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = blk.split()[0]
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("mpcx::engine::", "").replace("mpcx::models::", "").replace("mpcx::(anonymous namespace)::", "").replace("mpcx::", "")
    name = re.sub(r"Oscillators<(\d+)>", r"Oscillators<\1>", name)

    def g(key):
        m = re.search(key + r": (\d+)", blk)
        return int(m.group(1)) if m else -1
    if "__global__" in blk or True:
        print("%-62s VGPRs %3d AGPRs %3d  VGPR spill %3d  SGPR spill %3d  scratch %5d B/lane  occupancy %d waves/SIMD" %
              (name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
