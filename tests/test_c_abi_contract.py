"""The C ABI's argument contract (include/arrow_amd.h), exercised directly through ctypes on the
kernel sources built for the host (tests/emu): every entry point rejects NULL buffers, negative or
mismatched lengths, unsupported widths / types and short or misaligned workspaces with a status
code that mirrors arrow::StatusCode and a message in arx_last_error() — never a crash — and
zero-length inputs are accepted everywhere."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.emu

OK, INVALID, INDEX_ERROR, NOT_IMPLEMENTED = 0, -4, -7, -10


def _span(L, data=None, validity=None, offset=0, length=0, null_count=0):
    return L.ArxSpan(None if validity is None else validity.data_ptr(), None if data is None else data.data_ptr(),
                     offset, length, null_count)


def _buf(nbytes):
    return torch.zeros(max(nbytes, 64), dtype=torch.uint8)


def _err(lib):
    lib.arx_last_error.restype = C.c_char_p
    return lib.arx_last_error().decode()


def test_filter_and_take_argument_checks(emu_ctx):
    L = emu_ctx._lib
    lib = L.get_lib()
    n = 1000
    vals, mask, ws, out = _buf(n * 8), _buf(n // 8 + 8), _buf(lib.arx_filter_workspace_bytes(n) + 64), _buf(n * 8)
    m = _span(L, mask, length=n)
    v = _span(L, vals, length=n)
    cnt = C.c_int64(0)
    assert lib.arx_filter_count(C.byref(m), 0, ws.data_ptr(), 8, C.byref(cnt), None) == INVALID      # workspace too small
    assert "workspace" in _err(lib)
    assert lib.arx_filter_count(None, 0, ws.data_ptr(), ws.numel(), C.byref(cnt), None) == INVALID
    assert lib.arx_filter_count(C.byref(m), 7, ws.data_ptr(), ws.numel(), C.byref(cnt), None) == INVALID  # bad null selection
    assert lib.arx_filter_count(C.byref(m), 0, ws.data_ptr(), ws.numel(), C.byref(cnt), None) == OK and cnt.value == 0
    assert lib.arx_filter_exec(C.byref(v), 3, C.byref(m), 0, ws.data_ptr(), 0, out.data_ptr(), None, None) in (INVALID, NOT_IMPLEMENTED)  # width 3
    short = _span(L, vals, length=n - 1)
    assert lib.arx_filter_exec(C.byref(short), 8, C.byref(m), 0, ws.data_ptr(), 0, out.data_ptr(), None, None) == INVALID
    assert "same length" in _err(lib)
    idx = _buf(40)
    i = _span(L, idx, length=10)
    assert lib.arx_take(C.byref(v), 8, C.byref(i), 99, out.data_ptr(), None, None, None) == NOT_IMPLEMENTED  # index type
    assert lib.arx_take(C.byref(v), 5, C.byref(i), 5, out.data_ptr(), None, None, None) == NOT_IMPLEMENTED   # byte width
    assert lib.arx_take(C.byref(v), 8, C.byref(i), 5, None, None, None, None) == INVALID
    tws = _buf(lib.arx_take_workspace_bytes() + 64)
    idx.view(torch.int32)[3] = 5000
    assert lib.arx_check_index_bounds(C.byref(i), 5, 1000, tws.data_ptr(), tws.numel(), None) == INDEX_ERROR
    assert _err(lib) == "Index 5000 out of bounds"
    idx.view(torch.int32)[3] = -1
    assert lib.arx_check_index_bounds(C.byref(i), 5, 1000, tws.data_ptr(), tws.numel(), None) == INDEX_ERROR
    assert _err(lib) == "Index -1 out of bounds"
    empty = _span(L, idx, length=0)
    assert lib.arx_take(C.byref(v), 8, C.byref(empty), 5, out.data_ptr(), None, None, None) == OK


def test_sort_and_groupby_argument_checks(emu_ctx):
    L = emu_ctx._lib
    lib = L.get_lib()
    n = 500
    keys, out = _buf(n * 8), _buf(n * 8)
    need = lib.arx_sort_indices_workspace_bytes(n)
    ws = _buf(need + 512)
    base = (ws.data_ptr() + 255) & ~255
    v = _span(L, keys, length=n)
    assert lib.arx_sort_indices(C.byref(v), 0, 0, 1, base, 16, out.data_ptr(), None) == INVALID          # short workspace
    assert "workspace" in _err(lib)
    assert lib.arx_sort_indices(C.byref(v), 0, 0, 1, base + 8, need, out.data_ptr(), None) == INVALID     # misaligned
    assert lib.arx_sort_indices(C.byref(v), 42, 0, 1, base, need, out.data_ptr(), None) == NOT_IMPLEMENTED  # key type
    assert lib.arx_sort_indices(C.byref(v), 0, 9, 1, base, need, out.data_ptr(), None) == INVALID          # order
    assert lib.arx_sort_indices(C.byref(v), 0, 0, 9, base, need, out.data_ptr(), None) == INVALID          # null placement
    assert lib.arx_sort_indices(C.byref(v), 0, 0, 1, base, need, None, None) == INVALID
    assert lib.arx_sort_indices(C.byref(_span(L, keys, length=0)), 0, 0, 1, base, need, out.data_ptr(), None) == OK
    cap = 64
    state = _buf(lib.arx_groupby_state_bytes(cap))
    assert lib.arx_groupby_init(state.data_ptr(), 48, None) == INVALID                                     # not a power of two
    assert lib.arx_groupby_init(None, cap, None) == INVALID
    assert lib.arx_groupby_init(state.data_ptr(), cap, None) == OK
    k32, v64 = _buf(n * 4), _buf(n * 8)
    ks, vs = _span(L, k32, length=n), _span(L, v64, length=n - 3)
    assert lib.arx_groupby_sum_i64_consume(state.data_ptr(), cap, C.byref(ks), C.byref(vs), None, 0, None) == INVALID
    assert "same length" in _err(lib)
    # more distinct keys than slots: reported, not silently dropped
    k32.view(torch.int32)[:n] = torch.arange(n, dtype=torch.int32)
    vs = _span(L, v64, length=n)
    rc = lib.arx_groupby_sum_i64_consume(state.data_ptr(), cap, C.byref(ks), C.byref(vs), None, 0, None)
    g = C.c_int64(0)
    assert rc == INVALID or lib.arx_groupby_num_groups(state.data_ptr(), C.byref(g), None) == INVALID
    assert "capacity" in _err(lib) or "full" in _err(lib)
    mm = _buf(lib.arx_groupby_minmax_bytes(cap))
    assert lib.arx_groupby_minmax_i64_consume(state.data_ptr(), None, cap, C.byref(ks), C.byref(vs), None) == INVALID
    assert lib.arx_groupby_export(state.data_ptr(), None, k32.data_ptr(), k32.data_ptr(), v64.data_ptr(), v64.data_ptr(),
                                  k32.data_ptr(), v64.data_ptr(), v64.data_ptr(), None) == INVALID   # extrema wanted, no minmax buffer
    del mm
    # grouped float sums (row order) and rows / list elements of any width
    gids, sums, cnts, seen = _buf(n * 4), _buf(64 * 8), _buf(64 * 8), _buf(64 * 4)
    fv = _span(L, v64, length=n)
    fneed = lib.arx_hash_sum_float_workspace_bytes(n)
    fws = _buf(fneed + 512)
    args = (gids.data_ptr(), n, fws.data_ptr(), fneed + 512, sums.data_ptr(), cnts.data_ptr(), seen.data_ptr(), None)
    assert lib.arx_hash_sum_float_consume(C.byref(fv), 6, 0, 0.0, *args) == INVALID                       # int64 is not a float type
    assert lib.arx_hash_sum_float_consume(C.byref(fv), 9, 0, 0.0, gids.data_ptr(), n, fws.data_ptr(), 64, sums.data_ptr(), cnts.data_ptr(),
                                          seen.data_ptr(), None) == INVALID                             # short workspace
    assert "workspace" in _err(lib)
    assert lib.arx_hash_sum_float_consume(C.byref(fv), 9, 0, 0.0, None, n, fws.data_ptr(), fneed + 512, sums.data_ptr(), cnts.data_ptr(),
                                          seen.data_ptr(), None) == INVALID
    assert lib.arx_hash_sum_float_consume(C.byref(fv), 9, 0, 0.0, gids.data_ptr(), 0, None, 0, None, None, None, None) == OK
    assert lib.arx_hash_sum_float_consume(C.byref(fv), 9, 0, 0.0, *args) == OK                             # all rows in group 0
    assert lib.arx_hash_sum_float_workspace_bytes(0) == 0
    assert lib.arx_hash_sum_f64_merge(sums.data_ptr(), cnts.data_ptr(), seen.data_ptr(), None, cnts.data_ptr(), seen.data_ptr(),
                                      gids.data_ptr(), 4, None) == INVALID
    assert lib.arx_hash_sum_f64_merge(None, None, None, None, None, None, None, 0, None) == OK
    assert lib.arx_hash_mean_f64_finalize(sums.data_ptr(), None, 4, sums.data_ptr(), None) == INVALID
    assert lib.arx_hash_mean_f64_finalize(None, None, 0, None, None) == OK


def test_scalar_kernel_argument_checks(emu_ctx):
    L = emu_ctx._lib
    lib = L.get_lib()
    a, b, out = _buf(800), _buf(800), _buf(800)
    assert lib.arx_compare_f64(2, None, 0.0, None, 0.0, 100, out.data_ptr(), None) == INVALID          # scalar x scalar
    assert lib.arx_compare_i64(77, a.data_ptr(), 0, b.data_ptr(), 0, 100, out.data_ptr(), None) == INVALID   # unknown op
    assert lib.arx_compare_i64(2, a.data_ptr(), 0, b.data_ptr(), 0, 0, None, None) == OK
    assert lib.arx_arith_i64(9, a.data_ptr(), 0, b.data_ptr(), 0, 100, out.data_ptr(), None) == INVALID
    assert lib.arx_arith_checked_i64(0, a.data_ptr(), 0, None, 0, b.data_ptr(), 0, None, 0, 100, out.data_ptr(), None, None) == INVALID  # no flag
    assert lib.arx_cast_f64_f32(None, 10, out.data_ptr(), None) == INVALID
    sp = _span(L, a, length=100)
    assert lib.arx_cast_i64_i32(C.byref(sp), 0, None, 0, out.data_ptr(), None) == INVALID               # checked cast needs a workspace
    assert lib.arx_cast_i64_i32(C.byref(sp), 1, None, 0, out.data_ptr(), None) == OK                    # unchecked does not
    lm, rm = _span(L, a, length=100), _span(L, b, length=90)
    assert lib.arx_boolean_kleene(0, C.byref(lm), C.byref(rm), out.data_ptr(), None, None) == INVALID   # length mismatch
    rm = _span(L, b, validity=a, length=100, null_count=-1)
    assert lib.arx_boolean_kleene(0, C.byref(lm), C.byref(rm), out.data_ptr(), None, None) == INVALID   # nulls but no out validity
    assert lib.arx_boolean_kleene(5, C.byref(lm), C.byref(lm), out.data_ptr(), None, None) == INVALID
    assert lib.arx_set_option(b"no_such_option", 1) != 0
    bs = L.ArxBinarySpan(None, None, None, 0, 10, 0)
    i = _span(L, a, length=5)
    tot = C.c_int64(0)
    assert lib.arx_binary_take_offsets(C.byref(bs), C.byref(i), 5, None, 0, out.data_ptr(), None, None, C.byref(tot), None) == INVALID


def test_concatenate_and_delta_argument_checks(emu_ctx):
    L = emu_ctx._lib
    lib = L.get_lib()
    src, dst = _buf(64), _buf(64)
    assert lib.arx_bitmap_copy_at(src.data_ptr(), -1, 10, dst.data_ptr(), 0, None) == INVALID
    assert lib.arx_bitmap_copy_at(src.data_ptr(), 0, 10, None, 0, None) == INVALID
    assert lib.arx_bitmap_copy_at(src.data_ptr(), 0, 10, dst.data_ptr(), -3, None) == INVALID
    assert lib.arx_bitmap_copy_at(None, 0, 0, None, 0, None) == OK                                    # nothing to append
    assert lib.arx_bitmap_copy_at(None, 0, 70, dst.data_ptr(), 3, None) == OK                           # NULL source = all ones
    bits = np.unpackbits(dst.numpy()[:16], bitorder="little")
    assert bits[:3].sum() == 0 and bits[3:73].all() and bits[73:].sum() == 0
    assert lib.arx_binary_rebase_offsets(None, 4, 0, dst.data_ptr(), None) == INVALID
    assert lib.arx_binary_rebase_offsets(src.data_ptr(), -1, 0, dst.data_ptr(), None) == INVALID
    # DELTA_BINARY_PACKED: corrupt headers are refused by the host walk, never read past the page
    nmb, vpm, total, first, used = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_size_t(0)
    def scan(page):
        buf = np.frombuffer(page, dtype=np.uint8)
        return lib.arx_delta_scan_miniblocks(buf.ctypes.data, len(buf), 0, None, 0, C.byref(nmb), C.byref(vpm), C.byref(total),
                                             C.byref(first), C.byref(used))
    from oracle import oracle as O
    good = O.delta_binary_packed_encode(np.arange(1000) * 3)
    assert scan(good) == OK and (total.value, first.value, used.value) == (1000, 0, len(good))
    assert scan(good[: len(good) - 5]) == INVALID and "past the page" in _err(lib)                     # truncated miniblock
    assert scan(good[:2]) == INVALID                                                                  # truncated header
    assert scan(b"\x07\x04\x05\x00") == INVALID and "header" in _err(lib)                             # block size 7
    assert scan(b"\x80\x01\x00\x05\x00") == INVALID                                                   # zero miniblocks
    assert scan(b"\x80\x01\x04\x01\x02") == OK and (nmb.value, total.value, first.value) == (0, 1, 1)   # one value: no blocks
    assert lib.arx_delta_scan_miniblocks(None, 0, 0, None, 0, C.byref(nmb), C.byref(vpm), C.byref(total), C.byref(first), None) == INVALID
    out, ws = _buf(8000), _buf(lib.arx_delta_decode_workspace_bytes(1000) + 64)
    page = _buf(len(good) + 64)
    mbs = _buf(24 * 40)
    assert lib.arx_delta_decode(page.data_ptr(), mbs.data_ptr(), 32, 32, 0, 1000, 3, ws.data_ptr(), ws.numel(), out.data_ptr(), None) == INVALID  # width 3
    assert lib.arx_delta_decode(page.data_ptr() + 1, mbs.data_ptr(), 32, 32, 0, 1000, 8, ws.data_ptr(), ws.numel(), out.data_ptr(), None) == INVALID  # misaligned page
    assert lib.arx_delta_decode(page.data_ptr(), mbs.data_ptr(), 3, 32, 0, 1000, 8, ws.data_ptr(), ws.numel(), out.data_ptr(), None) == INVALID  # too few miniblocks
    assert lib.arx_delta_decode(page.data_ptr(), mbs.data_ptr(), 32, 32, 0, 1000, 8, ws.data_ptr(), 8, out.data_ptr(), None) == INVALID      # short workspace
    assert lib.arx_delta_decode(None, None, 0, 0, 0, 0, 8, None, 0, None, None) == OK                  # nothing to decode
    assert lib.arx_delta_decode(None, None, 0, 0, 42, 1, 8, ws.data_ptr(), ws.numel(), out.data_ptr(), None) == OK   # a single value: the header's
    assert out.view(torch.int64)[0].item() == 42
    # BYTE_STREAM_SPLIT
    assert lib.arx_byte_stream_split_decode(page.data_ptr(), 10, 3, out.data_ptr(), None) == NOT_IMPLEMENTED       # width 3
    assert lib.arx_byte_stream_split_decode(page.data_ptr(), -1, 4, out.data_ptr(), None) == INVALID
    assert lib.arx_byte_stream_split_decode(page.data_ptr(), 10, 8, out.data_ptr() + 4, None) == INVALID             # misaligned output
    assert lib.arx_byte_stream_split_decode(None, 10, 8, out.data_ptr(), None) == INVALID
    assert lib.arx_byte_stream_split_decode(None, 0, 8, None, None) == OK
    page[:16] = torch.arange(16, dtype=torch.uint8)
    assert lib.arx_byte_stream_split_decode(page.data_ptr(), 4, 4, out.data_ptr(), None) == OK
    assert out[:16].tolist() == [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
    # DELTA_LENGTH_BYTE_ARRAY: lengths -> offsets
    lens = torch.tensor([3, 0, 5, 1], dtype=torch.int32)
    offs = torch.zeros(16, dtype=torch.int32)
    assert lib.arx_lengths_to_offsets_i32(lens.data_ptr(), 4, 10, offs.data_ptr(), ws.data_ptr(), ws.numel(), None) == OK
    assert offs[:5].tolist() == [10, 13, 13, 18, 19]
    assert lib.arx_lengths_to_offsets_i32(None, 0, 7, offs.data_ptr(), ws.data_ptr(), ws.numel(), None) == OK and offs[0].item() == 7
    assert lib.arx_lengths_to_offsets_i32(None, 4, 0, offs.data_ptr(), ws.data_ptr(), ws.numel(), None) == INVALID
    assert lib.arx_lengths_to_offsets_i32(lens.data_ptr(), 4, 0, None, ws.data_ptr(), ws.numel(), None) == INVALID
    assert lib.arx_lengths_to_offsets_i32(lens.data_ptr(), 4, 0, offs.data_ptr(), ws.data_ptr(), 0, None) == INVALID
    assert lib.arx_lengths_to_offsets_i32(lens.data_ptr(), -1, 0, offs.data_ptr(), ws.data_ptr(), ws.numel(), None) == INVALID
    # DELTA_BYTE_ARRAY: ["apple", "apply", "", "b"] then a second page ["x", "xy"]
    prefix = torch.tensor([0, 4, 0, 0, 0, 1], dtype=torch.int32)
    slen = torch.tensor([5, 1, 0, 1, 1, 1], dtype=torch.int32)
    suffix = torch.tensor(list(b"appleybxy") + [0] * 7, dtype=torch.uint8)
    first = torch.tensor([0, 4, 6], dtype=torch.int64)
    sfirst = torch.tensor([0, 7, 9], dtype=torch.int64)
    vlen = torch.zeros(6, dtype=torch.int32)
    state = torch.zeros(2, dtype=torch.int64)
    soff, ooff = torch.zeros(7, dtype=torch.int32), torch.zeros(7, dtype=torch.int32)
    assert lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), 6, first.data_ptr(), 2, vlen.data_ptr(), state.data_ptr(), None) == OK
    assert vlen.tolist() == [5, 5, 0, 1, 1, 2] and state.tolist() == [0, 14]
    assert lib.arx_lengths_to_offsets_i32(slen.data_ptr(), 6, 0, soff.data_ptr(), ws.data_ptr(), ws.numel(), None) == OK
    assert lib.arx_lengths_to_offsets_i32(vlen.data_ptr(), 6, 100, ooff.data_ptr(), ws.data_ptr(), ws.numel(), None) == OK
    data = torch.zeros(16, dtype=torch.uint8)
    assert lib.arx_delta_byte_array_expand(prefix.data_ptr(), soff.data_ptr(), suffix.data_ptr(), 9, ooff.data_ptr(), 100, first.data_ptr(),
                                           sfirst.data_ptr(), 2, data.data_ptr(), state.data_ptr(), None) == OK
    assert bytes(data[:14].tolist()) == b"appleapplybxxy" and state[0].item() == 0
    sfirst[1] = 6                                              # the first page's suffix lengths add up to 7, not 6: both pages are off
    assert lib.arx_delta_byte_array_expand(prefix.data_ptr(), soff.data_ptr(), suffix.data_ptr(), 9, ooff.data_ptr(), 100, first.data_ptr(),
                                           sfirst.data_ptr(), 2, data.data_ptr(), state.data_ptr(), None) == OK
    assert state[0].item() == 16
    prefix[4] = 1                                              # a page's first value has nothing before it
    assert lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), 6, first.data_ptr(), 2, vlen.data_ptr(), state.data_ptr(), None) == OK
    assert state[0].item() == 2
    prefix[4], prefix[1] = 0, -1
    assert lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), 6, first.data_ptr(), 2, vlen.data_ptr(), state.data_ptr(), None) == OK
    assert state[0].item() == 1
    assert lib.arx_delta_byte_array_lengths(None, None, 0, None, 0, None, state.data_ptr(), None) == OK and state.tolist() == [0, 0]
    assert lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), 6, first.data_ptr(), 2, vlen.data_ptr(), None, None) == INVALID
    assert lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), 6, None, 2, vlen.data_ptr(), state.data_ptr(), None) == INVALID
    assert lib.arx_delta_byte_array_expand(None, None, None, 0, None, 0, None, None, 0, None, None, None) == OK
    assert lib.arx_delta_byte_array_expand(prefix.data_ptr(), soff.data_ptr(), suffix.data_ptr() + 1, 8, ooff.data_ptr(), 100, first.data_ptr(),
                                           None, 2, data.data_ptr(), None, None) == INVALID                   # misaligned suffix bytes
    assert lib.arx_delta_byte_array_expand(prefix.data_ptr(), soff.data_ptr(), suffix.data_ptr(), 9, ooff.data_ptr(), 100, first.data_ptr(),
                                           sfirst.data_ptr(), 2, data.data_ptr(), None, None) == INVALID      # a check without a state word
