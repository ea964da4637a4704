"""BASELINE.json's configurations at (or near) their full sizes on one MI355X, checked through
size-independent properties (the oracle cannot run billions of rows in seconds): subsequence /
round trip for filter+take, exact equality with an independent elementwise statement for
cast/compare, checksum + group count for hash_sum, sortedness + stability + permutation for
sort_indices.  torch is only the checker here (tests may use anything)."""
import pytest

pytestmark = pytest.mark.gpu


def _bits(t):
    import torch

    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=t.device)
    return (t.view(-1, 8).to(torch.uint8) * w).sum(dim=1, dtype=torch.uint8)


def _bits_chunked(n, p, gen, invert=False):
    """LSB-first bitmap of n (multiple of 8) Bernoulli(p) bits + the bool tensor, built in chunks."""
    import torch

    out = torch.empty(n // 8, dtype=torch.uint8, device="cuda")
    flags = torch.empty(n, dtype=torch.bool, device="cuda")
    step = 1 << 27
    for b in range(0, n, step):
        e = min(n, b + step)
        f = torch.rand(e - b, device="cuda", generator=gen) < p
        if invert:
            f = ~f
        flags[b:e] = f
        out[b // 8: e // 8] = _bits(f)
    return out, flags


def test_config2_filter_take_1b_rows(gpu_ctx):
    """configs[1]: Filter + Take on 1B-row int64 + validity bitmap, 10 % selectivity."""
    import torch

    amd = gpu_ctx
    n = 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(7)
    vals = torch.empty(n, dtype=torch.int64, device="cuda")
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        vals[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device="cuda", generator=g)
    mask_bits, sel = _bits_chunked(n, 0.1, g)
    valid_bits, valid = _bits_chunked(n, 0.1, g, invert=True)   # 10 % nulls
    values = amd.Array(amd.array.int64, n, [valid_bits, vals.view(torch.uint8)], -1, 0)
    mask = amd.Array(amd.array.bool_, n, [None, mask_bits], 0, 0)
    out = amd.compute.filter(values, mask)
    s = int(sel.sum())
    assert out.length == s
    want = vals[sel]
    assert torch.equal(out.data[: s * 8].view(torch.int64), want)            # the subsequence, in order
    want_valid = _bits(torch.cat([valid[sel], torch.zeros((-s) % 8, dtype=torch.bool, device="cuda")]))
    assert torch.equal(out.validity[: want_valid.numel()], want_valid)       # validity gathered bit-exactly
    idx = amd.compute.get_take_indices(mask)
    assert idx.length == s
    tk = amd.compute.take(values, idx, boundscheck=True)
    tv = _bits(torch.cat([valid[sel], torch.zeros((-s) % 8, dtype=torch.bool, device="cuda")]))
    assert torch.equal(tk.validity[: tv.numel()], tv)
    got_t = tk.data[: s * 8].view(torch.int64)
    vs = valid[sel]
    assert torch.equal(got_t[vs], want[vs]) and int(got_t[~vs].abs().sum()) == 0   # null slots zero-filled
    assert tk.null_count == s - int(vs.sum())


def test_config3_cast_and_greater_1b_rows(gpu_ctx):
    """configs[2]: cast float64->float32 and greater on 1B-row arrays; 0 ULP observed (the bar is 1)."""
    import torch

    amd = gpu_ctx
    n = 1_000_000_000
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        x[b:e] = torch.randn(e - b, dtype=torch.float64, device="cuda", generator=g)
        y[b:e] = torch.randn(e - b, dtype=torch.float64, device="cuda", generator=g)
    x[:1000] *= 1e300    # overflow to inf
    x[1000:2000] *= 1e-42  # float32 subnormals
    x[2000:2010] = float("nan")
    y[::7] = x[::7]      # ties
    ax = amd.Array(amd.array.float64, n, [None, x.view(torch.uint8)], 0, 0)
    ay = amd.Array(amd.array.float64, n, [None, y.view(torch.uint8)], 0, 0)
    c = amd.compute.cast(ax, amd.array.float32).data[: n * 4].view(torch.float32)
    for b in range(0, n, 1 << 28):
        e = min(n, b + (1 << 28))
        want = x[b:e].to(torch.float32)            # IEEE round-to-nearest-even, same as static_cast<float>
        assert torch.equal(c[b:e].view(torch.int32)[~want.isnan()], want.view(torch.int32)[~want.isnan()])
        assert bool(c[b:e][want.isnan()].isnan().all())
    gt = amd.compute.greater(ax, ay)
    for b in range(0, n, 1 << 28):
        e = min(n, b + (1 << 28))
        assert torch.equal(gt.data[b // 8: e // 8], _bits(x[b:e] > y[b:e]))


def test_config4_hash_sum_1b_rows_10m_keys(gpu_ctx):
    """configs[3] on one GPU's share (4B rows / 4): sums wrap, so  sum of group sums == sum of all
    values (mod 2^64), groups == distinct keys, and every group's sum == an independent scatter-add."""
    import torch

    amd = gpu_ctx
    n, groups = 1_000_000_000, 10_000_000
    g = torch.Generator(device="cuda").manual_seed(9)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    vals = torch.empty(n, dtype=torch.int64, device="cuda")
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        keys[b:e] = torch.randint(0, groups, (e - b,), dtype=torch.int32, device="cuda", generator=g)
        vals[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device="cuda", generator=g)
    kk = amd.Array(amd.array.int32, n, [None, keys.view(torch.uint8)], 0, 0)
    vv = amd.Array(amd.array.int64, n, [None, vals.view(torch.uint8)], 0, 0)
    lib = amd._lib.get_lib()
    lines0 = int(lib.arx_get_counter(b"groupby_slices_lines"))
    gk, gkv, gs, gvalid = amd.compute.group_by_sum(kk, vv, capacity=1 << 25)
    assert int(lib.arx_get_counter(b"groupby_slices_lines")) == lines0 + 1, "ids from [0, 1e7) must take the lines plan (round 6)"
    assert gk.numel() == groups and bool(gkv.all()) and bool(gvalid.all())
    assert int(gs.sum()) == int(vals.sum())
    want = torch.zeros(groups, dtype=torch.int64, device="cuda")
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        want.index_add_(0, keys[b:e].to(torch.int64), vals[b:e])
    assert torch.equal(gs, want[gk.to(torch.int64)])


def test_config5_sort_indices_1b_rows(gpu_ctx):
    """configs[4] on one GPU's share (2B rows / 2): a permutation, sorted, ties in row order."""
    import torch

    amd = gpu_ctx
    n = 1 << 30
    g = torch.Generator(device="cuda").manual_seed(10)
    k = torch.empty(n, dtype=torch.int64, device="cuda")
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        k[b:e] = torch.randint(-2**63, 2**63 - 1, (e - b,), dtype=torch.int64, device="cuda", generator=g)
    k[: 1 << 20] = k[(1 << 20): (1 << 21)]          # duplicates: stability is observable
    ak = amd.Array(amd.array.uint64, n, [None, k.view(torch.uint8)], 0, 0)
    idx = amd.compute.sort_indices(ak).data[: n * 8].view(torch.int64)
    seen = torch.zeros(n, dtype=torch.bool, device="cuda")
    seen[idx] = True
    assert bool(seen.all())                                          # a permutation of 0..n-1
    del seen
    ok = True
    prev_key, prev_idx = None, None
    for b in range(0, n, 1 << 27):
        e = min(n, b + (1 << 27))
        ii = idx[b:e]
        kk = k[ii]
        ku = kk ^ (-2**63)                                           # unsigned order via sign flip
        ok = ok and bool((ku[1:] >= ku[:-1]).all())
        ties = ku[1:] == ku[:-1]
        ok = ok and bool((ii[1:][ties] > ii[:-1][ties]).all())       # stable
        if prev_key is not None:
            ok = ok and (int(ku[0]) > prev_key or (int(ku[0]) == prev_key and int(ii[0]) > prev_idx))
        prev_key, prev_idx = int(ku[-1]), int(ii[-1])
    assert ok


def test_config5_sort_indices_2b_rows_single_gpu(gpu_ctx):
    """configs[4] at its full size on ONE GPU (the N=1 point of the 1/2/4/8 series), through the
    bench's own leg: the result is a permutation of 0..N-1 (count, sum and sum of squares of the
    row numbers) and the keys gathered through it are non-decreasing."""
    import torch

    import bench

    sec, rows, ok = bench.measure_sort(0, 1, torch.device("cuda", 0), 2_000_000_000, 1, 0)
    assert rows == 2_000_000_000 and ok
