#!/bin/bash
# Round-4 GPU session 6: concat_stats tuning: e2e + full-size tests, then the bench line twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s6; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_fullsize_parity_gpu.py tests/test_refexec_gpu.py -m gpu -q --timeout=900 -p no:cacheprovider > $O/pytest.log 2>&1
echo "== tests exit $?"; tail -n 4 $O/pytest.log
for E in a b; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_$E.log
  python3 -c "import json; d=json.load(open('$O/bench_$E.log')); print(d['value'], d['ms_per_step'], d['stages'])"
done
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/stats" -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu --no-profile --no-fp16 > "$OLDPWD/$O/stats.log" 2>&1)
F=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv && grep -E "concat|finalize|pgemm|rgb_conv" $O/kernel_stats.csv | cut -c1-140
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
