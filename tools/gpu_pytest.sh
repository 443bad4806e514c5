#!/bin/bash
# tools/gpu_pytest.sh <tag> <pytest args...> — run pytest -m gpu -v with the whole output kept in gpurun_out/<tag>.log; prints the
# summary line, every line that is not a PASSED one, and (after a crash) the last test that started.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out
cd $R
AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1} timeout ${STEP_TIMEOUT:-1200} python -X faulthandler -m pytest -m gpu -v "$@" > gpurun_out/$tag.log 2>&1
rc=$?
grep -v "PASSED\|^  File\|Extension modules\|amdgpu.ids" gpurun_out/$tag.log | grep -v "^$" | tail -${PY_TAIL:-40}
echo "[$tag] rc=$rc  passed=$(grep -c PASSED gpurun_out/$tag.log)  last started: $(grep -o '^tests/[^ ]*' gpurun_out/$tag.log | tail -1)"
