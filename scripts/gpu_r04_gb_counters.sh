#!/bin/bash
# Round 4: what the group-by's two passes wait on: SQ wave-cycle split, LDS activity / bank
# conflicts / atomics, and the L2's write-request mix for gbp_scatter_wide and gbp_aggregate at 4e9 rows / 1e7 keys;
# the same SQ split for the sort's three movers.  One PMC set per pass (kernel trace only beside it).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_e}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o "TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_EA0_RDREQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|SQ_LDS[A-Z0-9_]*\|SQ_INSTS_LDS[A-Z0-9_]*\|SQ_WAIT_INST_LDS\|SQ_INSTS_VMEM[A-Z0-9_]*\|TCP_[A-Z0-9_]*STALL[A-Z0-9_]*\|TCC_[A-Z0-9_]*STALL[A-Z0-9_]*\|SQ_INSTS_SALU\|SQ_INSTS_VALU\b" $OUT/counters_list.txt | sort -u | tr '\n' ' ' > $OUT/counter_names.txt
echo "names: $(cat $OUT/counter_names.txt | cut -c1-1500)"
pass() {   # name, what, counters...
  local name=$1 what=$2; shift 2
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/p_$name -o pmc -- python scripts/prof_sort_groupby.py $what 1 > /dev/null 2> $OUT/err_$name.txt
  echo "pass $name rc=$?"
  echo "== $*" >> $OUT/${what}_counters.txt
  python scripts/rocprof_summary.py pmc $(find $OUT/p_$name -name "*.db" | head -1) ${FLT:-gbp} 2>&1 | head -${HEAD:-14} >> $OUT/${what}_counters.txt
  find $OUT/p_$name -name "*.db" -delete
}
FLT=gbp pass gb_sq groupby SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
FLT=gbp pass gb_lds groupby SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU
FLT=gbp pass gb_tcc groupby TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
cat $OUT/groupby_counters.txt
FLT=msd HEAD=30 pass sort_sq sort SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
FLT=msd HEAD=16 pass sort_tcc sort TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
cat $OUT/sort_counters.txt
tail -3 $OUT/err_gb_lds.txt $OUT/err_gb_tcc.txt | cut -c1-300
