#!/bin/bash
# Round 4: striped rooms for the group-by's flat scatter (groupby_stripe) against room-after-room, end to end and by
# kernel trace; then the UTCL1 miss counters of the best stripe.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r04_x}
mkdir -p $OUT
export TMPDIR=/tmp
for lg in ${LGS:-0 8 6 10 12 14 0 8}; do
  echo "== groupby_stripe=$lg" | tee -a $OUT/stripe_ab.txt
  ARX_OPTIONS="groupby_stripe=$lg" timeout 200 python scripts/prof_sort_groupby.py groupby 2 2>&1 | grep "run [12]" | tee -a $OUT/stripe_ab.txt
done
for lg in ${TRACE_LGS:-0 8}; do
  ARX_OPTIONS="groupby_stripe=$lg" timeout 300 rocprofv3 --kernel-trace -d $OUT/t_$lg -o tr -- python scripts/prof_sort_groupby.py groupby 2 > /dev/null 2> $OUT/err_t_$lg.txt
  echo "== kernel trace, groupby_stripe=$lg" | tee -a $OUT/stripe_trace.txt
  python scripts/rocprof_summary.py trace $(find $OUT/t_$lg -name "*.db" | head -1) gbp 2>&1 | head -6 | tee -a $OUT/stripe_trace.txt
  find $OUT/t_$lg -name "*.db" -delete
done
lg=${PMC_LG:-276}
ARX_OPTIONS="groupby_stripe=$lg" timeout 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --kernel-trace -d $OUT/p_$lg -o pmc -- python scripts/prof_sort_groupby.py groupby 1 > /dev/null 2> $OUT/err_p_$lg.txt
echo "== UTCL1, groupby_stripe=$lg" | tee -a $OUT/stripe_trace.txt
python scripts/rocprof_summary.py pmc $(find $OUT/p_$lg -name "*.db" | head -1) gbp 2>&1 | head -9 | tee -a $OUT/stripe_trace.txt
find $OUT/p_$lg -name "*.db" -delete
