#!/bin/bash
# Same-box A/B: run CMD once per library variant, `rounds` times alternating.  usage: tools/ab_run.sh ROUNDS "lib1 lib2 ..." CMD...
# a library named "base" is the in-tree libltx2hip.so
rounds=$1; libs=$2; shift 2
for r in $(seq 1 $rounds); do
  for l in $libs; do
    if [ "$l" = base ]; then unset LTX2HIP_LIB; else export LTX2HIP_LIB=$PWD/ltx-2-mlx_amd/lib/ab/$l.so; fi
    echo "=== round $r lib $l"
    "$@"
  done
done
