#!/bin/bash
# Round 6 call R: rank / select_k / partition_nth on gfx950 (parity + the plugin script), the sort parity tests after the
# level-1 changes, and the rank kernels timed at 1e8 / 1e9 rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${RUN_TAG:-r06_r}
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_arrow_plugin.py tests/test_gpu_full_size.py -q -x -k "rank or select_k or sort" --durations=8 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest.log
timeout 600 python - > $OUT/rank_timing.txt 2>&1 <<'PY'
import time, torch, numpy as np
import arrow_amd as amd
dev = torch.device("cuda", 0)
for n in (100_000_000, 1_000_000_000):
    g = torch.Generator(device=dev).manual_seed(5)
    k = torch.empty(n, dtype=torch.int64, device=dev)
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        k[b:e] = torch.randint(0, n // 4, (e - b,), dtype=torch.int64, device=dev, generator=g)
    a = amd.Array(amd.array.int64, n, [None, k.view(torch.uint8)], 0, 0)
    for tb in ("first", "min", "dense", "quantile"):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = amd.compute.rank(a, "ascending", "at_end", tb) if tb != "quantile" else amd.compute.rank_quantile(a)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            del r
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = amd.compute.sort_indices(a); torch.cuda.synchronize(); sort_ms = (time.perf_counter() - t0) * 1e3
        del s
        print(f"rank[{tb}] int64 n={n}: {min(ts):.2f} ms (of which the sort {sort_ms:.2f} ms) = {n / min(ts) / 1e6:.2f} Grows/s", flush=True)
    del a, k
    torch.cuda.empty_cache()
PY
cat $OUT/rank_timing.txt
