#!/bin/bash
# Usage: bash tools/pmc.sh <tag> "<counters...>" <cmd...>   -> gpurun_out/<tag>/pmc.csv (kernel rows only)
TAG=$1; PMC=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o p -- "$@" > "$OUT/stdout.log" 2>&1
for f in $(find /tmp/pmc_$TAG -name "*counter_collection.csv"); do cp "$f" "$OUT/pmc.csv"; done
python - "$OUT/pmc.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "gemm" in k or "attn" in k or "conv" in k or "norm" in k:
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} last={v[-1]:.4g}")
PY
