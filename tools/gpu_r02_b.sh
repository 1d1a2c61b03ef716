#!/bin/bash
# round-2 GPU session B: all GPU tests (both element types), bench in both element types
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/parity_log.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "== pytest exit $?"; tail -n 60 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/bench.log 2>&1
echo "== bench exit $?"; tail -n 2 gpurun_out/bench.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --precision fp16 > gpurun_out/bench_fp16.log 2>&1
echo "== bench fp16 exit $?"; tail -n 2 gpurun_out/bench_fp16.log
