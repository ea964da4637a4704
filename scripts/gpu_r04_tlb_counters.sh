#!/bin/bash
# Round 4: are the flat scatters (group-by, sort level 1) held back by address translation or by DRAM credits?
# UTCL1 (per-CU TLB) requests / misses / stalls and the L2's EA credit stalls for gbp_scatter_wide / gbp_aggregate,
# the sort's three movers, and a plain streaming copy as the baseline.  One PMC set per pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_w}
mkdir -p $OUT
export TMPDIR=/tmp
pass() {   # name, what, filter, counters...
  local name=$1 what=$2 flt=$3; shift 3
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/p_$name -o pmc -- python scripts/prof_sort_groupby.py $what 1 ${ROWS:-} > /dev/null 2> $OUT/err_$name.txt
  echo "pass $name rc=$?"
  echo "== $what: $*" >> $OUT/tlb_counters.txt
  python scripts/rocprof_summary.py pmc $(find $OUT/p_$name -name "*.db" | head -1) $flt 2>&1 | head -${HEAD:-16} >> $OUT/tlb_counters.txt
  find $OUT/p_$name -name "*.db" -delete
}
for what in copy groupby sort; do
  case $what in copy) flt=copy;; groupby) flt=gbp;; sort) flt=msd;; esac
  pass ${what}_tlb $what $flt TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
  pass ${what}_tlbstall $what $flt TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum
  pass ${what}_ea $what $flt TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
done
cat $OUT/tlb_counters.txt
tail -2 $OUT/err_copy_tlb.txt $OUT/err_groupby_ea.txt | cut -c1-300
