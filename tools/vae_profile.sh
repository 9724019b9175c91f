#!/bin/bash
# Usage (GPU box, repo root): bash tools/vae_profile.sh <tag>  -- rocprofv3 kernel stats of decode_latent 768x512x65 (1 warm-up + 3 reps)
set -u
TAG=${1:-vae}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG -o vae -- python "$REPO/tools/vae_time.py" > "$OUT/stdout.log" 2>&1
for f in $(find /tmp/rp_$TAG -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; done
tail -3 "$OUT/stdout.log"
