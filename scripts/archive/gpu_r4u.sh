#!/bin/bash
# round 4, call u: the in-process protocol self-test / latency probe of the peer transport (one process per configuration, every
# device-side wait bounded by 2 s and every process by 90 s), then the process tests
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 SNS_PEER_TIMEOUT_MS=2000
L=gpurun_out/r4u_peer_selftest.log
: > $L
run() {
  echo "== $*" >> $L
  timeout -k 5 90 "$@" >> $L 2>&1 || echo "FAILED ($?)" >> $L
}
run python scripts/gpu_r4_peer_selftest.py 2 64 500
run python scripts/gpu_r4_peer_selftest.py 2 5776 500
run python scripts/gpu_r4_peer_selftest.py 3 64 500
run python scripts/gpu_r4_peer_selftest.py 3 5776 500
cat $L
unset SNS_PEER_TIMEOUT_MS
timeout -k 10 600 python -m pytest tests/test_gpu_peer.py -x -q -m gpu > gpurun_out/r4u_peer_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r4u_peer_tests.log
