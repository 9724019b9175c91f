#!/bin/bash
# Usage (GPU box, repo root): bash tools/pmc_bench.sh <tag> ["A B C"]  -> gpurun_out/<tag>/{pmcA,pmcB,pmcC}.csv + summary.md
# (bench.py runs with --no-graph: a TCC-counter pass over the hipGraph replay section did not finish in 30 minutes)
# Three separate --pmc passes of a short bench.py run (counters never combined with trace domains other than
# --kernel-trace): A = MFMA busy + GPU active cycles, B = FETCH_SIZE, C = WRITE_SIZE, D (on request: "A B C D") = L2 hit / miss and the
# fabric read requests with the share routed to the local memory controller (TCC_EA0_RDREQ_DRAM: Infinity-Cache hits are NOT told apart).
TAG=${1:-pmc_bench}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-loader --no-graph"
PASSES=${2:-"A B C"}
i=0
for PMC in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  n=$(echo A B C D | cut -d' ' -f$((i+1))); i=$((i+1))
  case " $PASSES " in *" $n "*) ;; *) continue;; esac
  rm -rf /tmp/pmcb_$n
  timeout 420 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmcb_$n -o p -- $CMD > "$OUT/stdout_$n.log" 2>&1
  for f in $(find /tmp/pmcb_$n -name "*counter_collection.csv"); do cp "$f" "$OUT/pmc$n.csv"; done
done
[ -f "$OUT/pmcA.csv" ] && python "$REPO/tools/pmc_summary.py" "$OUT/summary.md" $(ls "$OUT"/pmc[ABC].csv)
