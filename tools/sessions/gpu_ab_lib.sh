#!/bin/bash
# A/B of two library builds with the same kbench specs: tools/probe/base_lib/libgenpercept_hip.so (baseline) vs the in-tree one
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2; do
echo "== BASE (round $r)"; LD_LIBRARY_PATH=$PWD/tools/probe/base_lib timeout 200 tools/kbench "$@" | grep -vE "^#"
echo "== NEW (round $r)"; timeout 200 tools/kbench "$@" | grep -vE "^#"
done
