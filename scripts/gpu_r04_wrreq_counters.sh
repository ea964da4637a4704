#!/bin/bash
# Round 4: how many write requests the flat scatters send from the CUs to the L2 and how busy L2 / TA are — group-by and
# sort against a plain streaming copy.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_y}
mkdir -p $OUT
export TMPDIR=/tmp
pass() {   # name, what, filter, counters...
  local name=$1 what=$2 flt=$3; shift 3
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/p_$name -o pmc -- python scripts/prof_sort_groupby.py $what 1 > /dev/null 2> $OUT/err_$name.txt
  echo "pass $name rc=$?"
  echo "== $what: $*" >> $OUT/wrreq_counters.txt
  python scripts/rocprof_summary.py pmc $(find $OUT/p_$name -name "*.db" | head -1) $flt 2>&1 | head -${HEAD:-13} >> $OUT/wrreq_counters.txt
  find $OUT/p_$name -name "*.db" -delete
}
for what in ${WHATS:-copy groupby sort}; do
  case $what in copy) flt=copy;; groupby) flt=gbp;; sort) flt=msd;; esac
  pass ${what}_tcp $what $flt TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_WRITE_sum TCP_PENDING_STALL_CYCLES_sum
  pass ${what}_tcc $what $flt TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_TAG_STALL_sum
  pass ${what}_tccb $what $flt TCC_BUSY_sum TCC_CYCLE_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_sum
  pass ${what}_ta $what $flt TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum
done
cat $OUT/wrreq_counters.txt
