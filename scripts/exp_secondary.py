"""GPU experiment: the secondary configurations of SURVEY.md 8(d): take with RANDOM indices, filter
with 5 % mask nulls (DROP / EMIT_NULL), bounds-checked take.  1B-row int64 source, 1e8 outputs."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_amd as amd
import bench
dev = torch.device("cuda", 0)
n = 1_000_000_000
values, validity, mask = bench.gen_filter_inputs(n, dev, 77, 0.10, 0.10)
_, mvalid, _ = bench.gen_filter_inputs(1 << 20, dev, 78, 0.05, 0.5)
g = torch.Generator(device=dev).manual_seed(5)
mv = bench.pack_bits_device(torch.rand((n + 7) // 8 * 8, device=dev, generator=g) >= 0.05)
dv = amd.Array(amd.array.int64, n, [validity, values], -1, 0)
dm = amd.Array(amd.array.bool_, n, [None, mask], 0, 0)
dmn = amd.Array(amd.array.bool_, n, [mv, mask], -1, 0)
ridx = torch.randint(0, n, (100_000_000,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
di = amd.Array(amd.array.uint32, 100_000_000, [None, ridx.view(torch.uint8)], 0, 0)
mono = amd.compute.get_take_indices(dm)
out = {}
def t(name, fn, bytes_alg=None):
    ms = bench._time_gpu(fn, reps=5, warm=2)
    out[name] = {"ms": round(ms, 4)}
    if bytes_alg:
        out[name]["algorithmic_GBps"] = round(bytes_alg / ms / 1e6, 1)
    print(name, out[name], flush=True)
S = mono.length
t("filter DROP, mask without nulls (headline)", lambda: amd.compute.filter(dv, dm), 8*n + n/4 + 8.125*S)
t("filter DROP, 5% mask nulls", lambda: amd.compute.filter(dv, dmn), 8*n + 3*n/8 + 8.125*0.95*S)
t("filter EMIT_NULL, 5% mask nulls", lambda: amd.compute.filter(dv, dmn, "emit_null"), 8*n + 3*n/8 + 8.125*(0.95*S + 0.05*n))
t("take monotonic uint32[1e8], no boundscheck", lambda: amd.compute.take(dv, mono, boundscheck=False), 20.25*S)
t("take monotonic uint32[1e8], boundscheck", lambda: amd.compute.take(dv, mono, boundscheck=True), 20.25*S)
t("take RANDOM uint32[1e8], no boundscheck", lambda: amd.compute.take(dv, di, boundscheck=False), 20.25*1e8)
t("take RANDOM uint32[1e8], boundscheck", lambda: amd.compute.take(dv, di, boundscheck=True), 20.25*1e8)
print(json.dumps(out))
