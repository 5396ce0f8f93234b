cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "nuts" > gpurun_out/r6zg_nuts_tests.log 2>&1 < /dev/null; tail -2 gpurun_out/r6zg_nuts_tests.log
timeout 600 python tools/nuts_small_d.py 2>/dev/null < /dev/null | grep -v amdgpu > gpurun_out/r6zg_nuts_small_d.jsonl; python - <<PY
import json
for l in open("gpurun_out/r6zg_nuts_small_d.jsonl"):
    if l.startswith("{"):
        j=json.loads(l); print(j["target"], j["mode"], j["variant"], round(j["kernel_ms"],2))
PY
