#!/bin/bash
# Round-4 GPU session 17: (1) socket power and clocks (rocm-smi, twice a second) while the bench command runs back to back, next to the power cap: the
# direct reading behind "the pipeline runs at the chip's power limit"; (2) the same for one conv shape timed alone (kbench);
# (3) pipeline A/B of the tuned Winograd kernel (GENPERCEPT_WINO=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04s17; rm -rf $O; mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks > $O/smi_idle.txt 2>&1
sample() { while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> $1; echo >> $1; sleep 0.4; done; }
sample $O/smi_bench.jsonl & SMI=$!
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu --no-fp16 --no-profile 2>&1 | tail -1 > $O/bench_long.log
kill $SMI; wait $SMI 2>/dev/null
sample $O/smi_kbench.jsonl & SMI=$!
timeout 120 tools/kbench iters=400 cold=1 check=0 conv:4,384,384,256,256 | grep -vE "^#" > $O/kbench_long.log
kill $SMI; wait $SMI 2>/dev/null
for E in "default:" "wino:GENPERCEPT_WINO=1" "default2:" "wino2:GENPERCEPT_WINO=1"; do
  env ${E#*:} timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-fp16 2>&1 | tail -1 > $O/bench_${E%%:*}.log
  python3 -c "import json; d=json.load(open('$O/bench_${E%%:*}.log')); print('${E%%:*}', d['value'], d['ms_per_step'])"
done
python3 - <<'PY'
import json
O = "gpurun_out/r04s17"
print(open(O + "/smi_idle.txt").read()[:1500])
for f in ("smi_bench.jsonl", "smi_kbench.jsonl"):
    rows = []
    for l in open(f"{O}/{f}"):
        l = l.strip()
        if not l: continue
        try: rows.append(json.loads(l))
        except Exception: pass
    print(f, len(rows), "samples")
    if rows: print(json.dumps(rows[len(rows) // 2])[:600])
print(open(O + "/bench_long.log").read()[:200]); print(open(O + "/kbench_long.log").read()[:300])
PY
