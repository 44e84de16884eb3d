#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
T=${1:-r04g}
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${T}_smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/${T}_smoke.log | cut -c1-200
( timeout 1700 python -m pytest tests -m gpu -q -rA --timeout 600 2>&1 | grep -v "^PASSED" ) > $O/${T}_pytest.log 2>&1; tail -6 $O/${T}_pytest.log | cut -c1-220; grep "^parity\|^KKT" $O/${T}_pytest.log | cut -c1-300
( timeout 300 python bench.py ) > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err; python - <<PY
import json
d=json.loads(open("$O/${T}_bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","p50_step_latency_ms")}); print(json.dumps(d.get("nlmpc"),indent=0)[:1500])
PY
( MPCX_FORCE_DIST=1 timeout 200 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --pipeline-streams 0 --nlmpc-extra 0 ) 2> $O/${T}_bench_rccl1.err | grep "^{" > $O/${T}_bench_rccl1.json; python - <<PY
import json
d=json.loads(open("$O/${T}_bench_rccl1.json").read().strip().splitlines()[-1])
print("one-rank RCCL path: in series", d["value"], "overlapped", d["allgather_overlapped"])
PY
