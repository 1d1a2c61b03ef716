#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py "tests/test_e2e_gpu.py::test_modes_seg_matting_dis_on_the_gpu" -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_fullsize.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_r03a.log; cat gpurun_out/bench_r03a.log
