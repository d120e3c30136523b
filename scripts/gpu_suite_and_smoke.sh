#!/bin/bash
# round 5, call r5s: the whole GPU suite + the entry point's smoke on HEAD
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5s_gputests.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r5s_gputests.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r5s_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r5s_smoke.log | cut -c1-300
