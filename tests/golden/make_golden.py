"""Generates tests/golden/pyarrow_golden.npz by running the REFERENCE's own build (the pyarrow
25.0.0 wheel = libarrow.so.2500) on seeded inputs.  Commit the output; re-run only when the
fixture definition changes:  python tests/golden/make_golden.py"""
import os

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(0x0FF1CE)
n = 3000
out = {"n": n, "pyarrow_version": pa.__version__}
vals = rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)
valid = rng.random(n) >= 0.1
mask = rng.random(n) < 0.3
mvalid = rng.random(n) >= 0.05
out.update(values=vals, values_valid=valid, mask=mask, mask_valid=mvalid)
pv, pm = pa.array(vals, mask=~valid), pa.array(mask, mask=~mvalid)
for sel in ("drop", "emit_null"):
    r = pc.filter(pv, pm, null_selection_behavior=sel)
    out[f"filter_{sel}_valid"] = ~np.asarray(r.is_null())
    out[f"filter_{sel}_values"] = r.fill_null(0).to_numpy(zero_copy_only=False)
idx = rng.integers(0, n, size=1000).astype(np.int32)
ivalid = rng.random(1000) >= 0.1
r = pc.take(pv, pa.array(idx, mask=~ivalid))
out.update(indices=idx, indices_valid=ivalid, take_valid=~np.asarray(r.is_null()),
           take_values=r.fill_null(0).to_numpy(zero_copy_only=False))
f = rng.standard_normal(2000)
f[:6] = [0.0, -0.0, np.inf, -np.inf, 1e39, 1e-46]
f[100:300] *= 1e40
f[300:500] *= 1e-42
g = rng.standard_normal(2000)
g[::3] = f[::3]
out.update(f64=f, f64_b=g, cast_f32=pc.cast(pa.array(f), pa.float32(), safe=False).to_numpy(),
           greater=pc.greater(pa.array(f), pa.array(g)).to_numpy(zero_copy_only=False))
sk = rng.integers(0, 50, size=2500).astype(np.uint64)
sk[::2] = rng.integers(0, 2**63, size=len(sk[::2])).astype(np.uint64)
sv = rng.random(2500) >= 0.1
out.update(sort_keys=sk, sort_valid=sv)
for order in ("ascending", "descending"):
    for placement in ("at_end", "at_start"):
        out[f"sort_{order}_{placement}"] = pc.array_sort_indices(
            pa.array(sk, mask=~sv), order=order, null_placement=placement).to_numpy()
# array_sort_indices on the other key types: NaNs are null-likes next to the nulls, -0.0 ties 0.0
for name, dt in (("u32", np.uint32), ("i32", np.int32), ("f64", np.float64), ("f32", np.float32)):
    if np.dtype(dt).kind == "f":
        tk = rng.standard_normal(1500).astype(dt)
        tk[::7] = np.nan
        tk[::11] = 0.0
        tk[1::11] = -0.0
        tk[::13] = np.inf
        tk[5::13] = -np.inf
        tk[::3] = np.round(tk[::3])
    else:
        info = np.iinfo(dt)
        tk = rng.integers(info.min, info.max, size=1500, dtype=dt, endpoint=True)
        tk[::3] = tk[::3] % 9
    tv = rng.random(1500) >= 0.1
    out[f"tsort_{name}_keys"], out[f"tsort_{name}_valid"] = tk, tv
    for order in ("ascending", "descending"):
        for placement in ("at_end", "at_start"):
            out[f"tsort_{name}_{order}_{placement}"] = pc.array_sort_indices(
                pa.array(tk, mask=~tv), order=order, null_placement=placement).to_numpy()
gk = rng.integers(-30, 30, size=4000).astype(np.int32)
gkv = rng.random(4000) >= 0.03
gv = rng.integers(-2**63, 2**63 - 1, size=4000, dtype=np.int64)
gvv = rng.random(4000) >= 0.3
t = pa.table({"k": pa.array(gk, mask=~gkv), "v": pa.array(gv, mask=~gvv)})
r = t.group_by("k", use_threads=False).aggregate([("v", "sum")])
rk, rs = r.column("k").combine_chunks(), r.column("v_sum").combine_chunks()
out.update(gb_keys=gk, gb_keys_valid=gkv, gb_vals=gv, gb_vals_valid=gvv,
           gb_ref_isnull=np.asarray(rk.is_null()).astype(np.int64),
           gb_ref_key=rk.fill_null(0).to_numpy(zero_copy_only=False).astype(np.int64),
           gb_ref_valid=~np.asarray(rs.is_null()),
           gb_ref_sum=rs.fill_null(0).to_numpy(zero_copy_only=False))
np.savez_compressed(os.path.join(HERE, "pyarrow_golden.npz"), **out)
print("wrote pyarrow_golden.npz", {k: getattr(v, "shape", v) for k, v in out.items() if k != "n"})
