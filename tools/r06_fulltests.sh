#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3300 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_fulltests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1
tail -6 gpurun_out/r06_fulltests.log; tail -1 gpurun_out/r06_smoke.log
