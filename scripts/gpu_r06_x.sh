#!/bin/bash
# Round 6 call X: select_k by a threshold — parity on gfx950 (kernel tier + the plugin script), and its time beside the sort's at
# 1e9 / 2e9 rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_x}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py -q -x -k "select_k or rank_select" ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python - > $OUT/select_k_timing.txt 2>&1 <<'PY'
import time, torch
import arrow_amd as amd
dev = torch.device("cuda", 0)
for n in (1_000_000_000, 2_000_000_000):
    g = torch.Generator(device=dev).manual_seed(5)
    k_ = torch.empty(n, dtype=torch.int64, device=dev)
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        k_[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device=dev, generator=g)
    a = amd.Array(amd.array.int64, n, [None, k_.view(torch.uint8)], 0, 0)
    def timed(fn, reps=3):
        fn(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del r
        return min(ts)
    sort_ms = timed(lambda: amd.compute.sort_indices(a))
    for k in (10, 1000, 1_000_000, n // 20):
        c0 = dict(amd.compute._SELECT_COUNTERS)
        ms = timed(lambda: amd.compute.select_k_unstable(a, k, "descending"))
        took = {x: amd.compute._SELECT_COUNTERS[x] - c0[x] for x in c0}
        print(f"select_k_unstable int64 n={n} k={k}: {ms:.2f} ms ({took}); the sort alone {sort_ms:.2f} ms", flush=True)
    del a, k_
    torch.cuda.empty_cache()
PY
cat $OUT/select_k_timing.txt | grep -v amdgpu.ids
