#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash tools/rocprof_bench.sh <tag> [bench args...]
# Runs bench.py under `rocprofv3 --kernel-trace --stats` and leaves the per-kernel summary CSV plus
# bench.py's JSON line under gpurun_out/<tag>/ (copy what should be judged into profiles/).
set -u
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o bench -- python "$REPO/bench.py" "$@" > "$OUT/bench_stdout.log" 2>&1
grep '^{"metric"' "$OUT/bench_stdout.log" > "$OUT/bench.json" || true
for f in $(find /tmp/rp_$TAG -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
ls -la "$OUT"
head -25 "$OUT/kernel_stats.csv" 2>/dev/null | cut -c1-220
