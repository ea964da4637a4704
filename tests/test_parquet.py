"""Parquet column chunks decoded on the device (arrow_amd.parquet) against the reference's own
reader (pyarrow.parquet, i.e. cpp/src/parquet): every row group of every written variant —
dictionary / PLAIN encodings, data pages V1 / V2, several codecs, nulls, several pages per chunk —
must come out equal to `ParquetFile.read_row_group`.  The emulator runs are the CPU tier; the same
checks run on the GPU under `-m gpu`.  The run-header walk and the hybrid decode are also checked
against a numpy restatement (oracle/oracle.py::rle_hybrid_decode)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

from . import parity_cases as PC
from . import util as U
from .util import pa

pq = pytest.importorskip("pyarrow.parquet")


def _table(rng, n, null_p):
    m = (lambda: rng.random(n) < null_p) if null_p else (lambda: None)
    return pa.table({
        "i64_few": pa.array(rng.integers(-50, 50, n), mask=m()),                       # tiny dictionary, long runs
        "i64_wide": pa.array(rng.integers(-2**62, 2**62, n), mask=m()),               # big dictionary / PLAIN
        "i32_runs": pa.array(np.repeat(rng.integers(0, 9, n // 50 + 1), 50)[:n].astype(np.int32), mask=m()),
        "f64": pa.array(np.round(rng.standard_normal(n), 2), mask=m()),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m()),
        "req": pa.array(rng.integers(0, 1000, n)),
        "ts": pa.array(rng.integers(0, 2**50, n), pa.timestamp("us", tz="UTC"), mask=m()),      # logical types that share the
        "day": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m()),    # physical layout
        "tod": pa.array(rng.integers(0, 86_400_000, n).astype(np.int32), pa.time32("ms"), mask=m()),
        "flag": pa.array(rng.random(n) < 0.3, type=pa.bool_(), mask=m()),
        "flag_runs": pa.array(np.repeat(rng.random(n // 40 + 1) < 0.5, 40)[:n], type=pa.bool_(), mask=m()),
        "str": pa.array(np.array(["", "a", "bb", "gfx950", "MI355X", "ünïcödé", "x" * 40], dtype=object)[rng.integers(0, 7, n)], type=pa.string(), mask=m()),
    })


VARIANTS = [dict(compression="snappy", data_page_version="1.0", use_dictionary=True),
            dict(compression="none", data_page_version="1.0", use_dictionary=False),
            dict(compression="zstd", data_page_version="2.0", use_dictionary=True),
            dict(compression="none", data_page_version="2.0", use_dictionary=["str"], data_page_size=4096),
            dict(compression="gzip", data_page_version="1.0", use_dictionary=["i64_few", "i32_runs", "str"], data_page_size=2048)]


def check_file(amd, path):
    pf = pq.ParquetFile(path)
    got = amd.parquet.read_table(path)
    for name, chunks in got.items():
        assert len(chunks) == pf.metadata.num_row_groups
        for rg, arr in enumerate(chunks):
            want = pf.read_row_group(rg, columns=[name]).column(name).combine_chunks()
            have = arr.to_pyarrow()
            assert len(have) == len(want) and have.null_count == want.null_count == arr.null_count, (name, rg)
            assert have.equals(want), (name, rg, have.slice(0, 8), want.slice(0, 8))


def _write_and_check(amd, tmp_path, n, null_p, variant, seed):
    rng = np.random.default_rng(seed)
    path = os.path.join(tmp_path, "t.parquet")
    schema_nullable = _table(rng, n, null_p)
    fields = [pa.field(f.name, f.type, nullable=(f.name != "req")) for f in schema_nullable.schema]
    pq.write_table(schema_nullable.cast(pa.schema(fields)), path, row_group_size=max(1, n // 2 + 7), **variant)
    check_file(amd, path)


@pytest.mark.emu
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
@pytest.mark.parametrize("null_p", [0.0, 0.15])
def test_parquet_decode_emulator(emu_ctx, tmp_path, variant, null_p):
    _write_and_check(emu_ctx, str(tmp_path), 9000, null_p, VARIANTS[variant], 100 + variant)


SNAPPY_PAGE = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("dst_size", "<u4"), ("dst_offset", "<u8")])


def check_snappy_kernel(amd, rng, scale=1):
    """arx_snappy_decompress_pages vs the reference codec (pyarrow's SnappyCodec = the bundled snappy): blocks written
    by the reference compressor (empty, one byte, incompressible, periodic with every period length, long runs,
    integers), a hand-made block with a 4-byte-offset copy and a two-byte literal length (compressors rarely emit
    either), and corrupt blocks (offset beyond the output, truncated literal, wrong length) -> per-page status."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    codec = pa.Codec("snappy")

    def run(raws, blocks=None):
        blocks = blocks or [codec.compress(r).to_pybytes() for r in raws]
        pages = np.zeros(len(blocks), SNAPPY_PAGE)
        so = do = 0
        for i, (r, b) in enumerate(zip(raws, blocks)):
            pages[i] = (so, len(b), len(r), do)
            so += len(b)
            do += len(r)
        src = to_device(np.frombuffer(b"".join(blocks) + b"\0", dtype=np.uint8), dev)
        out = torch.zeros(max(do, 1) + 64, dtype=torch.uint8, device=dev)
        st = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        table = to_device(pages.view(np.uint8), dev)
        _lib.check(lib.arx_snappy_decompress_pages(src.data_ptr(), table.data_ptr(), len(blocks), out.data_ptr(),
                                                   st.data_ptr(), current_stream(dev)))
        return out.cpu().numpy()[:do].tobytes(), st.cpu().numpy().tolist(), out.cpu().numpy()[do:].tolist()

    raws = [b"", b"a", bytes(rng.integers(0, 256, 1000 * scale, dtype=np.uint8)), b"abcd" * 5000 * scale,
            bytes(rng.integers(0, 4, 70000 * scale, dtype=np.uint8)), np.arange(30000 * scale, dtype=np.int64).tobytes(),
            b"x" * 100000 * scale, (b"hello world, " * 300) + bytes(rng.integers(0, 256, 200, dtype=np.uint8))]
    raws += [bytes(rng.integers(0, 256, period, dtype=np.uint8)) * (3000 // period + 2) for period in (1, 2, 3, 5, 7, 13, 63, 64, 65, 200)]
    # copies that reach further back than the LDS ring holds (a 40 KB period), literals around the window size
    far = bytes(rng.integers(0, 256, 40_000, dtype=np.uint8))
    raws += [far * 3, bytes(rng.integers(0, 256, 30_000, dtype=np.uint8)) + bytes(1000) + far[:20_000] + far[:20_000]]
    raws += [bytes(rng.integers(0, 256, k, dtype=np.uint8)) + b"ab" * 40 for k in (1000, 1023, 1024, 1025, 2047, 2048, 2049, 3000)]
    # copies whose source sits exactly one LDS ring (16 KB) behind, i.e. in the slots the copy itself overwrites
    raws += [bytes(rng.integers(0, 256, p_, dtype=np.uint8)) * 3 for p_ in (16321, 16352, 16376, 16383, 16384, 16385, 16448)]
    # long literals (incompressible pages): the 16-bytes-per-lane copy with every head / tail length
    raws += [bytes(rng.integers(0, 256, k, dtype=np.uint8)) for k in (511, 512, 513, 527, 4097, 65536 + 3, 70001 * scale)]
    got, st, tail = run(raws)
    assert st == [0] * len(raws), st
    assert got == b"".join(raws) and not any(tail)
    lit = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    blk = bytes([0xB6, 0x02]) + bytes([61 << 2, 0x2B, 0x01]) + lit + bytes([(9 << 2) | 3, 0x2C, 0x01, 0, 0])
    got, st, _ = run([lit + lit[:10]], [blk])
    assert st == [0] and got == lit + lit[:10]
    bad1 = bytes([0x05]) + bytes([(4 << 2) | 2, 0x10, 0x00])          # copy from before the start of the output
    bad2 = bytes([0x0A]) + bytes([9 << 2]) + b"abc"                   # literal longer than the block
    bad3 = codec.compress(b"abcdef").to_pybytes()                     # announces 6 bytes, the page says 7
    _, st, _ = run([b"\0" * 5, b"\0" * 10, b"\0" * 7], [bad1, bad2, bad3])
    assert st == [3, 2, 1], st


def _with_snappy_form(amd, lds, fn):
    lib = amd._lib.get_lib()
    assert lib.arx_set_option(b"snappy_lds", lds) == 0
    try:
        fn()
    finally:
        lib.arx_set_option(b"snappy_lds", -1)


@pytest.mark.emu
@pytest.mark.parametrize("lds", [1, 0])
def test_snappy_page_decoder_kernel(emu_ctx, lds):
    """Both forms of the decoder: input window + recent output in LDS (default), and every byte through global memory."""
    _with_snappy_form(emu_ctx, lds, lambda: check_snappy_kernel(emu_ctx, np.random.default_rng(3)))


@pytest.mark.gpu
@pytest.mark.parametrize("lds", [1, 0])
def test_snappy_page_decoder_kernel_gpu(gpu_ctx, lds):
    _with_snappy_form(gpu_ctx, lds, lambda: check_snappy_kernel(gpu_ctx, np.random.default_rng(4), scale=20))


LZ4_BLOCK = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("stored", "<u4")])
LZ4_STREAM = np.dtype([("first_block", "<u8"), ("num_blocks", "<u4"), ("reserved", "<u4"), ("dst_offset", "<u8"), ("dst_size", "<u8")])


def lz4_scan_frames(lib, frames):
    """arx_lz4_frame_scan over several frames laid back to back: (concatenated bytes, blocks table, per-frame (first, n))."""
    import ctypes as C

    from arrow_amd import _lib

    blocks, spans, base = [], [], 0
    for f in frames:
        nb = C.c_int64(0)
        _lib.check(lib.arx_lz4_frame_scan(f, len(f), base, None, 0, C.byref(nb), None))
        tab = np.zeros(max(nb.value, 1), LZ4_BLOCK)
        _lib.check(lib.arx_lz4_frame_scan(f, len(f), base, tab.ctypes.data, nb.value, C.byref(nb), None))
        spans.append((len(blocks), nb.value))
        blocks.extend(tab[: nb.value].tolist())
        base += len(f)
    return b"".join(frames), np.array(blocks, LZ4_BLOCK) if blocks else np.zeros(1, LZ4_BLOCK), spans


def check_lz4_kernel(amd, rng, scale=1):
    """arx_lz4_frame_scan + arx_lz4_decompress_streams vs the reference codec (pyarrow's Lz4FrameCodec = the bundled lz4):
    frames written by the reference compressor — empty, tiny, incompressible (stored blocks), periodic with every
    period, long runs, buffers of several linked 64 KB blocks whose matches reach into the previous block — one stream
    per frame; corrupt input (a block cut short, a match offset before the start of the output, a wrong announced
    length) -> per-stream status."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    codec = pa.Codec("lz4")

    def run(raws, frames=None, sizes=None):
        frames = frames or [codec.compress(r).to_pybytes() for r in raws]
        sizes = sizes or [len(r) for r in raws]
        data, blocks, spans = lz4_scan_frames(lib, frames)
        streams = np.zeros(len(frames), LZ4_STREAM)
        do = 0
        for i, ((first, nb), n) in enumerate(zip(spans, sizes)):
            streams[i] = (first, nb, 0, do, n)
            do += n
        src = to_device(np.frombuffer(data + b"\0" * 8, dtype=np.uint8), dev)
        out = torch.zeros(max(do, 1) + 64, dtype=torch.uint8, device=dev)
        st = torch.full((len(frames),), 77, dtype=torch.int32, device=dev)
        d_streams, d_blocks = to_device(streams.view(np.uint8), dev), to_device(blocks.view(np.uint8), dev)
        _lib.check(lib.arx_lz4_decompress_streams(src.data_ptr(), d_streams.data_ptr(), d_blocks.data_ptr(), len(frames),
                                                  out.data_ptr(), st.data_ptr(), current_stream(dev)))
        o = out.cpu().numpy()
        return o[:do].tobytes(), st.cpu().numpy().tolist(), o[do:].tolist(), spans

    period = bytes(rng.integers(0, 256, 50_000, dtype=np.uint8))
    raws = [b"", b"a", b"ab" * 7, bytes(rng.integers(0, 256, 1000 * scale, dtype=np.uint8)), b"abcd" * 5000 * scale,
            bytes(rng.integers(0, 4, 70_000 * scale, dtype=np.uint8)), np.arange(30_000 * scale, dtype=np.int64).tobytes(),
            b"x" * 300_000 * scale, period * 5, np.cumsum(rng.integers(-3, 4, 40_000 * scale)).tobytes(),
            bytes(rng.integers(0, 256, 200_000, dtype=np.uint8))]
    raws += [bytes(rng.integers(0, 256, p_, dtype=np.uint8)) * (3000 // p_ + 2) for p_ in (1, 2, 3, 5, 7, 13, 63, 64, 65, 200)]
    got, st, tail, spans = run(raws)
    assert st == [0] * len(raws), st
    assert got == b"".join(raws) and not any(tail)
    assert max(n for _, n in spans) > 3          # several linked blocks in one frame
    # corrupt streams
    good = codec.compress(b"hello world, " * 200).to_pybytes()
    _, blocks, _ = lz4_scan_frames(lib, [good])
    cut = bytearray(good)
    cut_frame = bytes(cut[: int(blocks[0]["src_offset"]) + int(blocks[0]["src_size"]) // 2])      # (scan must reject it)
    import ctypes as C
    nb = C.c_int64(0)
    assert lib.arx_lz4_frame_scan(cut_frame, len(cut_frame), 0, None, 0, C.byref(nb), None) != 0
    assert lib.arx_lz4_frame_scan(b"\x00\x01\x02\x03\x04\x05\x06\x07", 8, 0, None, 0, C.byref(nb), None) != 0
    hdr = good[:7]

    def frame(block_bytes):
        return hdr + len(block_bytes).to_bytes(4, "little") + block_bytes + (0).to_bytes(4, "little")

    bad_offset = frame(bytes([0x10, 0x41, 0x05, 0x00]))               # 1 literal, then a match 5 bytes back
    bad_literal = frame(bytes([0xF0, 0x10]) + b"abc")                 # 31 literals announced, 3 there
    wrong_len = codec.compress(b"abcdef").to_pybytes()                # produces 6 bytes, the caller expects 7
    _, st, _, _ = run([b"\0" * 5, b"\0" * 40, b"\0" * 7], [bad_offset, bad_literal, wrong_len], [5, 40, 7])
    assert st == [3, 2, 1], st


@pytest.mark.emu
def test_lz4_stream_decoder_kernel(emu_ctx):
    check_lz4_kernel(emu_ctx, np.random.default_rng(7))


@pytest.mark.gpu
def test_lz4_stream_decoder_kernel_gpu(gpu_ctx):
    check_lz4_kernel(gpu_ctx, np.random.default_rng(8), scale=20)


LEVEL_PAGE = np.dtype([("byte_start", "<u8"), ("nbytes", "<u4"), ("num_values", "<u4"), ("row_start", "<u8")])


def check_levels_bitmap_kernel(amd, rng, scale=1):
    """arx_rle_levels_to_bitmap (run headers walked on the device, one wave per page) vs the levels themselves: blocks
    written by the hybrid encoder for random nulls at several densities, long runs, all-null and all-valid pages, pages
    of 1 / 7 / 8 / 9 / 504 / 505 values (partial groups; the 63-group limit of a literal run), pages that start at any
    row (neighbours share bitmap words), padding between the blocks; a hand-made block whose last literal run is cut
    short at the last needed byte; corrupt blocks (truncated, header past the block, a level of 2, a zero-length run)
    -> status 1 for that page only."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()

    def run(level_arrays, blocks=None, counts=None):
        blocks = blocks or [O.rle_hybrid_encode(v, 1) for v in level_arrays]
        counts = counts or [len(v) for v in level_arrays]
        pages = np.zeros(len(blocks), LEVEL_PAGE)
        buf, row = bytearray(), 0
        for i, (b, c) in enumerate(zip(blocks, counts)):
            buf += bytes(rng.integers(0, 256, int(rng.integers(0, 5)), dtype=np.uint8))   # (blocks sit anywhere)
            pages[i] = (len(buf), len(b), c, row)
            buf += b
            row += c
        src = to_device(np.frombuffer(bytes(buf) + b"\0" * 8, dtype=np.uint8), dev)
        bits = torch.zeros((row + 63) // 64 + 2, dtype=torch.int64, device=dev)
        ones = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        st = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        table = to_device(pages.view(np.uint8), dev)
        _lib.check(lib.arx_rle_levels_to_bitmap(src.data_ptr(), table.data_ptr(), len(blocks), bits.data_ptr(),
                                                ones.data_ptr(), st.data_ptr(), current_stream(dev)))
        got = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")
        return got, ones.cpu().numpy().tolist(), st.cpu().numpy().tolist(), row

    n = 3000 * scale
    levels = [(rng.random(n) >= p).astype(np.uint8) for p in (0.1, 0.5, 0.9, 0.01)]
    levels += [np.ones(n, np.uint8), np.zeros(n, np.uint8), np.repeat(rng.integers(0, 2, n // 37 + 1), 37)[:n].astype(np.uint8)]
    levels += [(rng.random(k) >= 0.3).astype(np.uint8) for k in (1, 7, 8, 9, 63, 64, 65, 504, 505, 511, 513, 4097)]
    order = rng.permutation(len(levels))
    levels = [levels[i] for i in order]
    got, ones, st, rows = run(levels)
    want = np.concatenate(levels)
    assert st == [0] * len(levels), st
    assert ones == [int(v.sum()) for v in levels]
    assert (got[:rows] == want).all() and not got[rows:].any()
    # the last literal run of a block may stop at the last byte its needed values use (RleBitPackedDecoder reads no further)
    v = (rng.random(20) >= 0.4).astype(np.uint8)
    full = O.rle_hybrid_encode(v, 1)          # one literal run of 3 groups: header + 3 bytes
    assert len(full) == 4
    got, ones, st, rows = run([v, v], [full, full], [20, 17])   # 17 values need 3 bytes as well
    assert st == [0, 0] and ones == [int(v.sum()), int(v[:17].sum())]
    assert (got[:37] == np.concatenate([v, v[:17]])).all() and not got[37:].any()
    # corrupt blocks
    good = (rng.random(100) >= 0.2).astype(np.uint8)
    enc = O.rle_hybrid_encode(good, 1)
    bad_trunc = enc[: len(enc) // 2]                   # ends before all its values
    bad_header = bytes([0x80, 0x80, 0x80])             # a varint that never ends inside the block
    bad_level = bytes([100 << 1, 2])                   # repeated run of the level 2
    bad_zero = bytes([0x00, 0x01])                     # a repeated run of zero values
    bad_groups = bytes([(60 << 1) | 1, 0xFF])          # a literal run far longer than the block
    got, ones, st, rows = run([good] * 7, [enc, bad_trunc, bad_header, enc, bad_level, bad_zero, bad_groups], [100] * 7)
    assert st == [0, 1, 1, 0, 1, 1, 1], st
    assert ones[0] == ones[3] == int(good.sum())
    assert (got[:100] == good).all() and (got[300:400] == good).all()


@pytest.mark.emu
def test_levels_bitmap_kernel_under_random_damage(emu_ctx):
    """Damaged level blocks: a status per page, bits only inside the page's own row range (the neighbours' bits and the
    rows behind the last page stay as the undamaged blocks set them / zero)."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    rng = np.random.default_rng(51)
    levels = [(rng.random(k) >= p).astype(np.uint8) for k, p in ((700, 0.1), (513, 0.5), (64, 0.3), (2000, 0.02), (9, 0.5))]
    good = [O.rle_hybrid_encode(v, 1) for v in levels]
    for trial in range(60):
        blocks, damaged = [], []
        for b in good:
            b = bytearray(b)
            hit = rng.random() < 0.6
            if hit:
                for _ in range(int(rng.integers(1, 4))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            blocks.append(bytes(b))
            damaged.append(hit and bytes(b) != good[len(blocks) - 1])
        pages = np.zeros(len(blocks), LEVEL_PAGE)
        buf, row = bytearray(), 0
        for i, (b, v) in enumerate(zip(blocks, levels)):
            pages[i] = (len(buf), len(b), len(v), row)
            buf += b
            row += len(v)
        src = to_device(np.frombuffer(bytes(buf) + b"\0" * 8, dtype=np.uint8), dev)
        bits = torch.zeros((row + 63) // 64 + 4, dtype=torch.int64, device=dev)
        ones = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        st = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        table = to_device(pages.view(np.uint8), dev)
        _lib.check(lib.arx_rle_levels_to_bitmap(src.data_ptr(), table.data_ptr(), len(blocks), bits.data_ptr(),
                                                ones.data_ptr(), st.data_ptr(), current_stream(dev)))
        got = np.unpackbits(bits.cpu().numpy().view(np.uint8), bitorder="little")
        status = st.cpu().numpy().tolist()
        at = 0
        for i, v in enumerate(levels):
            assert status[i] in (0, 1)
            if not damaged[i]:
                assert status[i] == 0 and (got[at: at + len(v)] == v).all(), (trial, i)
            at += len(v)
        assert not got[row:].any()


@pytest.mark.emu
def test_levels_bitmap_kernel(emu_ctx):
    check_levels_bitmap_kernel(emu_ctx, np.random.default_rng(5))


@pytest.mark.gpu
def test_levels_bitmap_kernel_gpu(gpu_ctx):
    check_levels_bitmap_kernel(gpu_ctx, np.random.default_rng(6), scale=40)


def _snappy_plain_file(tmp_path, n, null_p, version):
    rng = np.random.default_rng(n + int(null_p * 100))
    mask = (rng.random(n) < null_p) if null_p else None
    t = pa.table({"opt": pa.array(np.cumsum(rng.integers(-3, 4, n)), mask=mask),          # compressible int64
                  "req": pa.array(rng.integers(0, 50, n).astype(np.int32)),
                  "dbl": pa.array(np.round(rng.standard_normal(n), 1), mask=mask)})
    fields = [pa.field(f.name, f.type, nullable=(f.name != "req")) for f in t.schema]
    path = os.path.join(tmp_path, "snappy.parquet")
    pq.write_table(t.cast(pa.schema(fields)), path, use_dictionary=False, compression="snappy", data_page_version=version,
                   data_page_size=8192, row_group_size=n // 2 + 3)
    return path


@pytest.mark.emu
@pytest.mark.parametrize("version,null_p", [("2.0", 0.0), ("2.0", 0.2), ("1.0", 0.0), ("1.0", 0.2)])
def test_parquet_snappy_pages_decompressed_on_the_device(emu_ctx, tmp_path, version, null_p):
    """PLAIN fixed-width pages under Snappy never touch a host codec: V2 page bodies (levels travel uncompressed) and V1
    pages of required columns go through arx_snappy_decompress_pages; optional V1 pages (levels inside the block) keep
    the host codec.  Equal to pyarrow.parquet either way; the stats say how many pages took the device route."""
    path = _snappy_plain_file(str(tmp_path), 20000, null_p, version)
    stats = {}
    emu_ctx.parquet.read_table(path, stats=stats)
    want_device = {"2.0": 3, "1.0": 1}[version]       # columns whose pages qualify
    assert stats.get("device_snappy_pages", 0) >= 2 * want_device, stats
    check_file(emu_ctx, path)
    emu_ctx.parquet.DEVICE_SNAPPY = False
    try:
        check_file(emu_ctx, path)                      # the host-codec route stays covered
    finally:
        emu_ctx.parquet.DEVICE_SNAPPY = True


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["2.0", "1.0"])
def test_parquet_snappy_pages_decompressed_on_the_device_gpu(gpu_ctx, tmp_path, version):
    path = _snappy_plain_file(str(tmp_path), 600_000, 0.1, version)
    stats = {}
    gpu_ctx.parquet.read_table(path, stats=stats)
    assert stats.get("device_snappy_pages", 0) > 10, stats
    check_file(gpu_ctx, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.2])
def test_parquet_dictionary_fallback_to_plain(emu_ctx, tmp_path, null_p):
    """A chunk whose dictionary outgrows its page: the first data pages are dictionary-encoded, the
    rest PLAIN (what a 300k-distinct-value int64 column does at the default 1 MiB limit)."""
    rng = np.random.default_rng(9)
    n = 12_000
    t = pa.table({"wide": pa.array(rng.integers(-2**62, 2**62, n), mask=(rng.random(n) < null_p) if null_p else None),
                  "few": pa.array(rng.integers(0, 7, n))})
    path = os.path.join(str(tmp_path), "fb.parquet")
    pq.write_table(t, path, dictionary_pagesize_limit=8192, data_page_size=4096, compression="snappy")
    col = pq.ParquetFile(path).metadata.row_group(0).column(0)
    assert "PLAIN" in col.encodings and "RLE_DICTIONARY" in col.encodings
    check_file(emu_ctx, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.25])
def test_parquet_plain_and_fallback_strings(emu_ctx, tmp_path, null_p):
    """PLAIN byte-array pages (length-prefixed values), alone and after a dictionary -> PLAIN fallback:
    the page bytes become alternating {prefix, value} entries of the take's source array."""
    rng = np.random.default_rng(13)
    n = 9000
    words = np.array([("w%d" % i) * (i % 5) for i in range(n)], dtype=object)        # many distinct values, some empty
    mask = (rng.random(n) < null_p) if null_p else None
    t = pa.table({"s": pa.array(words[rng.integers(0, n, n)], type=pa.string(), mask=mask),
                  "b": pa.array([bytes([i % 251]) * (i % 7) for i in range(n)], type=pa.binary(), mask=mask)})
    plain = os.path.join(str(tmp_path), "plain.parquet")
    pq.write_table(t, plain, use_dictionary=False, data_page_size=2048, compression="snappy", row_group_size=5000)
    check_file(emu_ctx, plain)
    fb = os.path.join(str(tmp_path), "fallback.parquet")
    pq.write_table(t, fb, dictionary_pagesize_limit=4096, data_page_size=2048, compression="zstd", data_page_version="2.0")
    enc = pq.ParquetFile(fb).metadata.row_group(0).column(0).encodings
    assert "PLAIN" in enc and "RLE_DICTIONARY" in enc
    check_file(emu_ctx, fb)


@pytest.mark.emu
def test_parquet_growing_dictionary_widens_the_indices(emu_ctx, tmp_path):
    """New values keep arriving, so every page's indices are written with a wider bit width than the
    page before: one launch decodes runs of different widths."""
    n = 30_000
    t = pa.table({"grow": pa.array(np.arange(n) // 4, mask=np.arange(n) % 11 == 0), "one": pa.array(np.zeros(n, dtype=np.int64))})
    path = os.path.join(str(tmp_path), "grow.parquet")
    pq.write_table(t, path, data_page_size=1024, compression="none")
    check_file(emu_ctx, path)


@pytest.mark.emu
def test_parquet_edge_cases_emulator(emu_ctx, tmp_path):
    rng = np.random.default_rng(5)
    for n, null_p in ((0, 0.0), (1, 0.0), (1, 1.0), (70, 1.0), (5000, 0.999)):
        _write_and_check(emu_ctx, str(tmp_path), n, null_p, VARIANTS[0], n)
    with pytest.raises(emu_ctx.ArrowNotImplementedError):     # a struct with a list inside is out of scope
        path = os.path.join(str(tmp_path), "l.parquet")
        pq.write_table(pa.table({"l": pa.array([{"a": [1]}, None])}), path)
        emu_ctx.parquet.read_table(path)
    # list columns at the edges: no row, one null list, one empty list, only null / only empty lists, one long list
    lt = pa.list_(pa.field("element", pa.int64()))
    for k, values in enumerate(([], [None], [[]], [None] * 70, [[]] * 70, [list(range(9000))], [[None] * 130, None, [], [7]])):
        path = os.path.join(str(tmp_path), f"edge{k}.parquet")
        pq.write_table(pa.table({"l": pa.array(values, lt)}), path)
        got = emu_ctx.parquet.read_table(path)["l.list.element"]
        want = pq.read_table(path).column("l")
        assert len(got) == want.num_chunks or (len(values) == 0 and len(got) <= 1)
        if got:
            have = got[0].to_pyarrow()
            have.validate(full=True)
            assert have.equals(want.combine_chunks()), (values[:3], have)
    path = os.path.join(str(tmp_path), "b.parquet")             # binary (not utf8) values, some empty, some null
    pq.write_table(pa.table({"b": pa.array([b"\x00\x01", None, b"", b"\xff" * 9, None, b"\x00\x01"], pa.binary())}), path)
    check_file(emu_ctx, path)
    del rng


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_decode_gpu(gpu_ctx, tmp_path, variant, null_p):
    # (600k full-range int64 values also exercise the dictionary -> PLAIN fallback at the 1 MiB limit)
    _write_and_check(gpu_ctx, str(tmp_path), 600_000, null_p, VARIANTS[variant], 200 + variant)


def _delta_table(rng, n, null_p):
    mask = (lambda: rng.random(n) < null_p) if null_p else (lambda: None)
    walk = np.cumsum(rng.integers(-50, 60, n))                                   # small deltas: narrow miniblocks
    return pa.table({
        "sorted64": pa.array(np.sort(rng.integers(0, 2**40, n)), mask=mask()),
        "walk32": pa.array(walk.astype(np.int32), mask=mask()),
        "full64": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=mask()),       # deltas wrap around: 64-bit miniblocks
        "const32": pa.array(np.full(n, 7, dtype=np.int32), mask=mask()),           # all deltas equal: bit width 0
        "ticks": pa.array(np.arange(n) * 1000 + rng.integers(0, 3, n), mask=mask()),                  # timestamp-like: near-constant deltas
        "step64": pa.array(np.where(np.arange(n) % 700 == 0, 2**45, 3).cumsum(), mask=mask()),
    })


def _write_delta_and_check(amd, tmp_path, n, null_p, seed, **kw):
    rng = np.random.default_rng(seed)
    t = _delta_table(rng, n, null_p)
    path = os.path.join(tmp_path, "delta.parquet")
    pq.write_table(t, path, use_dictionary=False, column_encoding={name: "DELTA_BINARY_PACKED" for name in t.schema.names},
                   row_group_size=max(1, n // 2 + 7), **kw)
    md = pq.ParquetFile(path).metadata
    assert all("DELTA_BINARY_PACKED" in md.row_group(0).column(i).encodings for i in range(md.num_columns))
    check_file(amd, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.2])
@pytest.mark.parametrize("n,kw", [(9000, dict(data_page_size=2048, compression="snappy")),
                                  (4097, dict(data_page_version="2.0", compression="none")), (1, {}), (130, {})])
def test_parquet_delta_binary_packed_emulator(emu_ctx, tmp_path, n, kw, null_p):
    """DELTA_BINARY_PACKED INT32 / INT64 pages (DeltaBitPackDecoder): header walk on the host, unpack + prefix
    sum in the kernels; sorted, random-walk, wrap-around, constant and bursty columns, many small pages."""
    _write_delta_and_check(emu_ctx, str(tmp_path), n, null_p, 31 + n, **kw)




def _write_split_and_check(amd, tmp_path, n, null_p, seed, **kw):
    rng = np.random.default_rng(seed)
    mask = (lambda: rng.random(n) < null_p) if null_p else (lambda: None)
    t = pa.table({"f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=mask()),
                  "f64": pa.array(rng.standard_normal(n) * 1e100, mask=mask()),
                  "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32), mask=mask()),
                  "i64": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=mask())})
    path = os.path.join(tmp_path, "bss.parquet")
    pq.write_table(t, path, use_dictionary=False, column_encoding={name: "BYTE_STREAM_SPLIT" for name in t.schema.names},
                   row_group_size=max(1, n // 2 + 7), **kw)
    md = pq.ParquetFile(path).metadata
    assert all("BYTE_STREAM_SPLIT" in md.row_group(0).column(i).encodings for i in range(md.num_columns))
    check_file(amd, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.2])
@pytest.mark.parametrize("n,kw", [(9000, dict(data_page_size=2048, compression="snappy")), (1, {}),
                                  (4097, dict(data_page_version="2.0", compression="none"))])
def test_parquet_byte_stream_split_emulator(emu_ctx, tmp_path, n, kw, null_p):
    """BYTE_STREAM_SPLIT pages (ByteStreamSplitDecoder): float, double, int32, int64 columns, many small pages."""
    _write_split_and_check(emu_ctx, str(tmp_path), n, null_p, 51 + n, **kw)



def _write_delta_length_and_check(amd, tmp_path, n, null_p, seed, **kw):
    rng = np.random.default_rng(seed)
    mask = (lambda: rng.random(n) < null_p) if null_p else (lambda: None)
    words = np.array([("w%d" % i) * (i % 5) for i in range(max(n, 1))], dtype=object)
    t = pa.table({"s": pa.array(words[rng.integers(0, max(n, 1), n)], type=pa.string(), mask=mask()),
                  "b": pa.array([bytes([i % 251]) * (i % 9) for i in range(n)], type=pa.binary(), mask=mask()),
                  "empty": pa.array([""] * n, type=pa.string(), mask=mask()),
                  "long": pa.array(["x" * int(k) for k in rng.integers(0, 300, n)], type=pa.string(), mask=mask())})
    path = os.path.join(tmp_path, "dl.parquet")
    pq.write_table(t, path, use_dictionary=False, column_encoding={name: "DELTA_LENGTH_BYTE_ARRAY" for name in t.schema.names},
                   row_group_size=max(1, n // 2 + 7), **kw)
    md = pq.ParquetFile(path).metadata
    assert all("DELTA_LENGTH_BYTE_ARRAY" in md.row_group(0).column(i).encodings for i in range(md.num_columns))
    check_file(amd, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.2])
@pytest.mark.parametrize("n,kw", [(6000, dict(data_page_size=2048, compression="snappy")), (1, {}), (33, {}),
                                  (4097, dict(data_page_version="2.0", compression="none"))])
def test_parquet_delta_length_byte_array_emulator(emu_ctx, tmp_path, n, kw, null_p):
    """DELTA_LENGTH_BYTE_ARRAY pages (DeltaLengthByteArrayDecoder): delta-packed lengths -> offsets by a device prefix
    sum, bytes appended as they are; utf8 and binary, empty strings, long values, many small pages, nulls."""
    _write_delta_length_and_check(emu_ctx, str(tmp_path), n, null_p, 61 + n, **kw)



# ---- DELTA_BYTE_ARRAY (DeltaByteArrayDecoderImpl, parquet/decoder.cc:1974-2204)
def _dba_words(rng, n, kind):
    if kind == "sorted":          # what the encoding is for: sorted keys sharing long prefixes
        return sorted(("key/%08d/%s" % (int(k), "x" * int(k % 7))).encode() for k in rng.integers(0, 10 * max(n, 1), n))
    if kind == "random":          # no shared prefixes at all, empty values in between
        return [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) if k else b"" for k in rng.integers(0, 40, n)]
    if kind == "repeat":          # whole values repeated (prefix = the whole previous value, empty suffix), and shrinking
        base = [b"", b"a", b"ab", b"abc", b"abcd" * 9, b"abcd" * 9 + b"e"]
        return [base[int(i)] for i in rng.integers(0, len(base), n)]
    long_ = bytes(rng.integers(97, 123, 20000, dtype=np.uint8))      # values longer than the kernel's 8 KB LDS window
    return [long_[: int(k)] + bytes([65 + int(k) % 26]) for k in rng.choice([10, 5000, 8191, 8192, 8193, 12000, 19999], n)]


def _dba_pages(path, column):
    """(value bytes, number of non-null values) of every DELTA_BYTE_ARRAY data page of row group 0 (required columns)."""
    from arrow_amd import parquet as P

    col = pq.ParquetFile(path).metadata.row_group(0).column(column)
    raw = open(path, "rb").read()
    pages = []
    for hdr, payload in P._column_chunk_pages(raw, col):
        if hdr[1] == P._PAGE_DATA:
            assert hdr[5][2] == P._ENC_DELTA_BYTE_ARRAY
            pages.append((bytes(P._decompress(col.compression, payload, hdr[2])), hdr[5][1]))
    return pages


def test_delta_byte_array_oracle_pinned_to_the_reference_writer(tmp_path):
    """The restatement decodes the pages the reference's DeltaByteArrayEncoder wrote to pyarrow's own values (every page
    starts from the empty string), and refuses what the reference's decoder refuses."""
    rng = np.random.default_rng(5)
    n = 4000
    cols = {k: _dba_words(rng, n, k) for k in ("sorted", "random", "repeat")}
    cols["long"] = _dba_words(rng, 40, "long") * 100
    path = os.path.join(str(tmp_path), "pin.parquet")
    schema = pa.schema([pa.field(k, pa.binary(), nullable=False) for k in cols])
    pq.write_table(pa.table({k: pa.array(v, pa.binary()) for k, v in cols.items()}).cast(schema), path, use_dictionary=False,
                   compression="none", column_encoding={k: "DELTA_BYTE_ARRAY" for k in cols}, data_page_size=4096)
    for ci, (name, want) in enumerate(cols.items()):
        pages = _dba_pages(path, ci)
        assert len(pages) > 1, name
        got = []
        for page, count in pages:
            vals = O.delta_byte_array_decode(page)
            assert len(vals) == count
            got += vals
        assert got == want, name
    page = O.delta_byte_array_encode([b"abc", b"abd"])
    assert O.delta_byte_array_decode(page) == [b"abc", b"abd"]
    bad = O.delta_binary_packed_encode(np.array([1, 0], dtype=np.int64)) + O.delta_binary_packed_encode(np.array([2, 1], dtype=np.int64)) + b"xyz"
    with pytest.raises(ValueError, match="prefix length too large"):      # the first value of a page has nothing before it
        O.delta_byte_array_decode(bad)
    bad = O.delta_binary_packed_encode(np.array([0, -1], dtype=np.int64)) + O.delta_binary_packed_encode(np.array([2, 1], dtype=np.int64)) + b"xyz"
    with pytest.raises(ValueError, match="negative prefix length"):
        O.delta_byte_array_decode(bad)


def _check_dba_kernels(amd):
    """arx_delta_byte_array_lengths / _expand (through arrow_amd.parquet.decode_delta_byte_array) against the restatement:
    several pages in one launch, counts around the 64-value groups, values around the 8 KB LDS window, suffix groups
    larger than the staging buffer, block shapes the reference writer never produces — and the decoder's errors."""
    from arrow_amd import _lib

    rng = np.random.default_rng(17)
    for kind, counts, shape in (("sorted", (1, 63, 64, 65, 129, 1000), (128, 4)), ("random", (2, 64, 300), (256, 8)),
                                ("repeat", (5, 128, 777), (128, 1)), ("long", (3, 70), (128, 4))):
        pages, want = [], []
        for c in counts:
            vals = _dba_words(rng, c, kind)
            pages.append((O.delta_byte_array_encode(vals, *shape), c))
            assert O.delta_byte_array_decode(pages[-1][0]) == vals
            want += vals
        pages.insert(1, (b"", 0))                       # a page without values (all nulls) carries nothing
        arr = amd.parquet.decode_delta_byte_array(pages)
        assert arr.to_pyarrow().to_pylist() == want, kind
    two = lambda p, s, tail: (O.delta_binary_packed_encode(np.array(p, dtype=np.int64)) +
                              O.delta_binary_packed_encode(np.array(s, dtype=np.int64)) + tail)
    for page, text in ((two([1, 0], [2, 1], b"xyz"), "prefix length too large"), (two([0, 3], [2, 1], b"xyz"), "prefix length too large"),
                       (two([0, -1], [2, 1], b"xyz"), "negative prefix length"), (two([0, 1], [2, 5], b"xyz"), "do not add up")):
        with pytest.raises(_lib.ArrowInvalid, match=text):
            amd.parquet.decode_delta_byte_array([(O.delta_byte_array_encode([b"ok", b"okay"]), 2), (page, 2)])


@pytest.mark.emu
def test_delta_byte_array_kernels_vs_restatement(emu_ctx):
    _check_dba_kernels(emu_ctx)


@pytest.mark.emu
@pytest.mark.parametrize("seed", [11, 12])
def test_delta_byte_array_corrupt_pages_are_rejected_or_decoded_like_the_restatement(emu_ctx, seed):
    """Pages with 1 - 3 flipped bytes, some cut short: the device route never decodes what the restatement refuses and never
    decodes differently — and survives them (the emulated buffers are host memory: an out-of-bounds lane corrupts the heap.
    A negative prefix length once made the expansion write before a value's start; corrupt lengths now stop before it)."""
    rng = np.random.default_rng(seed)
    accepted = 0
    for it in range(150):
        n = int(rng.integers(1, 300))
        vals = sorted(bytes(rng.integers(97, 100, int(rng.integers(0, 12)), dtype=np.uint8)) for _ in range(n))
        page = bytearray(O.delta_byte_array_encode(vals))
        for _ in range(int(rng.integers(1, 4))):
            page[int(rng.integers(0, len(page)))] = int(rng.integers(0, 256))
        if it % 7 == 0:
            page = page[: int(rng.integers(1, len(page) + 1))]
        try:
            want = O.delta_byte_array_decode(bytes(page))
            want = want if len(want) == n else None
        except Exception:
            want = None
        try:
            got = emu_ctx.parquet.decode_delta_byte_array([(bytes(page), n)]).to_pyarrow().to_pylist()
        except Exception:
            got = None
        assert got is None or got == want, (seed, it)
        accepted += got is not None
    assert accepted > 30


def _write_dba_and_check(amd, tmp_path, n, null_p, seed, **kw):
    rng = np.random.default_rng(seed)
    mask = (lambda: rng.random(n) < null_p) if null_p else (lambda: None)
    t = pa.table({"sorted": pa.array([w.decode() for w in _dba_words(rng, n, "sorted")], pa.string(), mask=mask()),
                  "random": pa.array(_dba_words(rng, n, "random"), pa.binary(), mask=mask()),
                  "repeat": pa.array([w.decode() for w in _dba_words(rng, n, "repeat")], pa.string(), mask=mask()),
                  "long": pa.array((_dba_words(rng, 7, "long") * (n // 7 + 1))[:n], pa.binary(), mask=mask())})
    path = os.path.join(tmp_path, "dba.parquet")
    pq.write_table(t, path, use_dictionary=False, column_encoding={name: "DELTA_BYTE_ARRAY" for name in t.schema.names},
                   row_group_size=max(1, n // 2 + 7), **kw)
    md = pq.ParquetFile(path).metadata
    assert all("DELTA_BYTE_ARRAY" in md.row_group(0).column(i).encodings for i in range(md.num_columns))
    check_file(amd, path)


@pytest.mark.emu
@pytest.mark.parametrize("null_p", [0.0, 0.2])
@pytest.mark.parametrize("n,kw", [(3000, dict(data_page_size=2048, compression="snappy")), (1, {}), (65, {}),
                                  (2049, dict(data_page_version="2.0", compression="none"))])
def test_parquet_delta_byte_array_emulator(emu_ctx, tmp_path, n, kw, null_p):
    """DELTA_BYTE_ARRAY column chunks (DeltaByteArrayDecoderImpl) end to end: sorted keys with long shared prefixes, values
    without any, repeated values, values longer than the LDS window; utf8 and binary, many small pages, V1 / V2, nulls."""
    _write_dba_and_check(emu_ctx, str(tmp_path), n, null_p, 91 + n, **kw)


def _delta_page_bytes(path, column):
    """The value bytes of every DELTA_BINARY_PACKED data page of one column chunk, read with this package's page walk."""
    from arrow_amd import parquet as P

    md = pq.ParquetFile(path).metadata
    col = md.row_group(0).column(column)
    raw = open(path, "rb").read()
    pages = []
    for hdr, payload in P._column_chunk_pages(raw, col):
        if hdr[1] == P._PAGE_DATA:
            assert hdr[5][2] == P._ENC_DELTA_BINARY_PACKED
            pages.append((bytes(P._decompress(col.compression, payload, hdr[2])), hdr[5][1]))
    return pages


def test_delta_binary_packed_oracle_pinned_to_the_reference_writer(tmp_path):
    """The restatement decodes the pages the reference's DeltaBitPackEncoder wrote to pyarrow's own values, and the
    host header walk (arx_delta_scan_miniblocks) agrees with it on counts, first value and bytes consumed."""
    from arrow_amd import parquet as P

    rng = np.random.default_rng(3)
    n = 3000
    cols = {"a": np.sort(rng.integers(0, 2**40, n)), "b": rng.integers(-2**63, 2**63 - 1, n), "c": np.full(n, -5, dtype=np.int64),
            "d": np.cumsum(rng.integers(-9, 9, n)).astype(np.int32)}
    path = os.path.join(str(tmp_path), "pin.parquet")
    required = pa.schema([pa.field(k, pa.from_numpy_dtype(v.dtype), nullable=False) for k, v in cols.items()])   # no level block
    pq.write_table(pa.table({k: pa.array(v) for k, v in cols.items()}).cast(required), path, use_dictionary=False, compression="none",
                   column_encoding={k: "DELTA_BINARY_PACKED" for k in cols}, data_page_size=4096)
    for ci, (name, want) in enumerate(cols.items()):
        got = []
        for page, count in _delta_page_bytes(path, ci):
            vals, used = O.delta_binary_packed_decode(page)
            assert len(vals) == count
            mbs, vpm, total, first, consumed = P.scan_delta_miniblocks(page)
            assert (total, first, consumed) == (count, int(vals[0]), used) and vpm % 32 == 0
            assert len(mbs) == -(-(count - 1) // vpm)
            got.append(vals)
        got = np.concatenate(got)
        assert np.array_equal(got.astype(want.dtype), want), name


def _varint(x):
    out = bytearray()
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


def test_corrupt_run_and_block_headers_are_rejected_not_wrapped():
    """Crafted headers whose byte counts would wrap around size_t (a 2^62-group literal run, a 2^62-value block with
    one miniblock) are refused — DeltaBitPackDecoder reads its header fields as uint32 and RleBitPackedDecoder
    bounds every bit read — and a literal run may only be short where it reaches the last value."""
    from arrow_amd import _lib
    from arrow_amd import parquet as P

    # DELTA_BINARY_PACKED: block_size = 2^62, 1 miniblock per block, 1000 values, first = 0
    evil = _varint(1 << 62) + _varint(1) + _varint(1000) + _varint(0) + _varint(0) + bytes([1]) + bytes(64)
    with pytest.raises(_lib.ArrowInvalid):
        P.scan_delta_miniblocks(evil)
    evil = _varint(128) + _varint(1 << 40) + _varint(1000) + _varint(0)
    with pytest.raises(_lib.ArrowInvalid):
        P.scan_delta_miniblocks(evil)
    # RLE hybrid, bit width 3: a literal run announcing 2^62 groups
    with pytest.raises(_lib.ArrowInvalid):
        P.scan_rle_runs(_varint(((1 << 62) << 1) | 1) + bytes(16), 3, 100)
    # a literal run in the MIDDLE of the block that is cut short must fail even though 8 slack bytes would cover it
    block = _varint((2 << 1) | 1) + bytes(4) + _varint(10 << 1) + bytes([1])      # 16 values need 6 bytes, 4 present
    with pytest.raises(_lib.ArrowInvalid):
        P.scan_rle_runs(block, 3, 26)
    # ... while the LAST run may stop where its needed values stop: 2 groups announced, 9 values needed = 4 bytes
    runs, _ = P.scan_rle_runs(_varint((2 << 1) | 1) + bytes(4), 3, 9)
    assert len(runs) == 1


def test_delta_last_miniblock_need_not_be_padded():
    """A final miniblock holding fewer values than its size may be cut after the bytes those values use (the
    reference decodes only what it needs); the walk accepts it and the restatement decodes the same values."""
    from arrow_amd import parquet as P

    vals = np.arange(0, 40 * 7, 7, dtype=np.int64) ** 2          # 40 values: one block, 39 deltas in miniblocks 0 and 1
    page = O.delta_binary_packed_encode(vals, 128, 4)
    mbs, vpm, total, first, used = P.scan_delta_miniblocks(page)
    assert (vpm, total, used) == (32, 40, len(page)) and len(mbs) == 2
    bw = int(mbs[1]["bit_width"]) if "bit_width" in mbs.dtype.names else None
    # cut the padding of the last miniblock: 7 deltas live in it
    if bw:
        keep = len(page) - (32 * bw // 8) + (7 * bw + 7) // 8
        mbs2, vpm2, total2, first2, used2 = P.scan_delta_miniblocks(page[:keep])
        assert (vpm2, total2, first2, used2) == (vpm, total, first, keep) and len(mbs2) == 2
        with pytest.raises(Exception):
            P.scan_delta_miniblocks(page[:keep - 1])


@pytest.mark.emu
@pytest.mark.parametrize("block_size,miniblocks", [(128, 4), (256, 8), (128, 1), (1024, 4)])
def test_delta_decode_kernel_vs_restatement(emu_ctx, block_size, miniblocks):
    """arx_delta_decode against the restatement on pages of block shapes the reference writer never produces
    (the format allows any multiple of 128 with miniblocks of a multiple of 32 values), INT32 and INT64 output."""
    PC.check_delta_decode(emu_ctx, np.random.default_rng(block_size + miniblocks), block_size, miniblocks)


def test_rle_run_walk_and_decode_vs_numpy_restatement():
    """scan_rle_runs + the oracle's hybrid decode reproduce what the reference's encoder wrote:
    definition levels and dictionary indices of a written file decode to pyarrow's own values."""
    from arrow_amd import parquet as P

    rng = np.random.default_rng(11)
    for bit_width in (1, 3, 7, 12, 20):
        n = 5000
        vals = rng.integers(0, 1 << bit_width, n).astype(np.uint32)
        vals[100:900] = vals[100]                      # a long repeated run among literals
        enc = O.rle_hybrid_encode(vals, bit_width)
        runs, ones = P.scan_rle_runs(enc, bit_width, n)
        got = O.rle_hybrid_decode(enc, runs, bit_width, n)
        assert (got == vals).all()
        if bit_width == 1:
            assert ones == int(vals.sum())


@pytest.mark.emu
def _check_ipc(amd, tmp_path, n):
    rng = np.random.default_rng(3)
    t = _table(rng, n, 0.1)
    for compression in (None, "lz4", "zstd"):
        path = os.path.join(str(tmp_path), f"t_{compression}.arrow")
        opts = pa.ipc.IpcWriteOptions(compression=compression)
        with pa.ipc.new_file(path, t.schema, options=opts) as w:
            for b in t.to_batches(max_chunksize=n // 3 + 1):
                w.write_batch(b)
        stats = {}
        got = amd.ipc.read_table(path, stats=stats, device_decompress=True)
        # LZ4_FRAME bodies: every buffer decompressed on the device (arx_lz4_decompress_streams); ZSTD and plain
        # bodies through the reference's reader
        assert stats["device_lz4_batches"] == (3 if compression == "lz4" else 0), (compression, stats)
        ref = pa.ipc.open_file(pa.memory_map(path, "r"))
        assert set(got) == set(t.schema.names)
        for name, chunks in got.items():
            assert len(chunks) == ref.num_record_batches
            for i, arr in enumerate(chunks):
                want = ref.get_batch(i).column(name)
                assert arr.to_pyarrow().equals(want) and arr.null_count == want.null_count, (name, i)
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t, max_chunksize=n // 2)
    got = amd.ipc.read_table(pa.BufferReader(sink.getvalue()), columns=["i64_few", "str"])
    assert [a.length for a in got["str"]] == [n // 2, n // 2]
    assert pa.chunked_array([a.to_pyarrow() for a in got["i64_few"]]).equals(t.column("i64_few"))


def _check_ipc_corrupt_lz4_body(amd, tmp_path):
    """A garbled LZ4 block inside an IPC body is reported (per-stream status -> OSError, the class pyarrow raises for
    'LZ4 decompress failed'), an implausible uncompressed length is refused before anything is allocated."""
    import struct

    from arrow_amd import ipc as I

    t = pa.table({"a": pa.array(np.arange(20_000, dtype=np.int64) // 7)})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_file(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression="lz4")) as w:
        w.write_table(t)
    raw = bytearray(sink.getvalue().to_pybytes())
    msgs = [m for m in pa.ipc.MessageReader.open_stream(pa.BufferReader(bytes(raw[8:]))) if m.type == "record batch"]
    info = I.parse_record_batch_message(msgs[0].metadata.to_pybytes())
    assert info["codec"] == 0 and info["length"] == 20_000 and info["nodes"] == [(20_000, 0)]
    body = msgs[0].body.to_pybytes()
    at = raw.find(body[:64])
    off, length = info["buffers"][1]
    bad = bytearray(raw)
    for k in range(at + off + 8 + 11 + 6, at + off + 8 + 11 + 40):      # inside the first block's sequences
        bad[k] ^= 0xFF
    try:
        got = amd.ipc.read_table(pa.BufferReader(bytes(bad)), device_decompress=True)
        # (LZ4 has no checksum here: garbage may decode to other bytes of the right length — then they differ)
        assert not got["a"][0].to_pyarrow().equals(t.column("a").chunk(0))
    except (OSError, amd._lib.ArrowInvalid):
        pass
    huge = bytearray(raw)
    huge[at + off: at + off + 8] = struct.pack("<q", 1 << 50)
    with pytest.raises(amd._lib.ArrowInvalid):
        amd.ipc.read_table(pa.BufferReader(bytes(huge)), device_decompress=True)


@pytest.mark.emu
@pytest.mark.parametrize("lds", [1, 0])
def test_snappy_decoder_under_random_damage(emu_ctx, lds):
    """Both decoder forms on blocks with random byte damage: every page ends with a status, a damaged page never
    writes outside its own destination range (the bytes between and behind the ranges stay zero), undamaged pages of
    the same launch still decode."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    def body():
        lib, dev = _lib.get_lib(), default_device()
        rng = np.random.default_rng(31 + lds)
        codec = pa.Codec("snappy")
        raws = [bytes(rng.integers(0, 5, 3000, dtype=np.uint8)), b"abcdefgh" * 500, np.arange(700, dtype=np.int64).tobytes(),
                bytes(rng.integers(0, 256, 2500, dtype=np.uint8))]
        good = [codec.compress(r).to_pybytes() for r in raws]
        for trial in range(60):
            blocks = []
            for b in good:
                b = bytearray(b)
                if rng.random() < 0.7:
                    for _ in range(int(rng.integers(1, 5))):
                        b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
                blocks.append(bytes(b))
            gap = 96
            pages = np.zeros(len(blocks), SNAPPY_PAGE)
            so = do = 0
            for i, (r, b) in enumerate(zip(raws, blocks)):
                pages[i] = (so, len(b), len(r), do)
                so += len(b)
                do += len(r) + gap
            src = to_device(np.frombuffer(b"".join(blocks) + b"\0" * 8, dtype=np.uint8), dev)
            out = torch.zeros(do + 64, dtype=torch.uint8, device=dev)
            st = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
            table = to_device(pages.view(np.uint8), dev)
            _lib.check(lib.arx_snappy_decompress_pages(src.data_ptr(), table.data_ptr(), len(blocks), out.data_ptr(),
                                                       st.data_ptr(), current_stream(dev)))
            o, status = out.cpu().numpy(), st.cpu().numpy().tolist()
            at = 0
            for i, r in enumerate(raws):
                assert status[i] in (0, 1, 2, 3), status
                if blocks[i] == good[i]:
                    assert status[i] == 0 and o[at: at + len(r)].tobytes() == r
                assert not o[at + len(r): at + len(r) + gap].any(), (trial, i, "wrote behind its range")
                at += len(r) + gap

    _with_snappy_form(emu_ctx, lds, body)


@pytest.mark.emu
def test_lz4_decoder_under_random_damage(emu_ctx):
    """The same for the LZ4 stream decoder: damaged block bytes (the frame structure intact) never write outside the
    stream's destination range."""
    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()
    rng = np.random.default_rng(41)
    codec = pa.Codec("lz4")
    raws = [bytes(rng.integers(0, 5, 3000, dtype=np.uint8)), b"abcdefgh" * 500, np.arange(700, dtype=np.int64).tobytes(),
            bytes(rng.integers(0, 4, 90_000, dtype=np.uint8))]
    good = [codec.compress(r).to_pybytes() for r in raws]
    for trial in range(40):
        frames = []
        for f in good:
            _, blocks, _ = lz4_scan_frames(lib, [f])
            f = bytearray(f)
            if rng.random() < 0.7:
                b = blocks[int(rng.integers(0, len(blocks)))]
                for _ in range(int(rng.integers(1, 5))):
                    f[int(b["src_offset"]) + int(rng.integers(0, int(b["src_size"])))] = int(rng.integers(0, 256))
            frames.append(bytes(f))
        data, blocks, spans = lz4_scan_frames(lib, frames)
        gap = 96
        streams = np.zeros(len(frames), LZ4_STREAM)
        do = 0
        for i, ((first, nb), r) in enumerate(zip(spans, raws)):
            streams[i] = (first, nb, 0, do, len(r))
            do += len(r) + gap
        src = to_device(np.frombuffer(data + b"\0" * 8, dtype=np.uint8), dev)
        out = torch.zeros(do + 64, dtype=torch.uint8, device=dev)
        st = torch.full((len(frames),), 77, dtype=torch.int32, device=dev)
        d_streams, d_blocks = to_device(streams.view(np.uint8), dev), to_device(blocks.view(np.uint8), dev)
        _lib.check(lib.arx_lz4_decompress_streams(src.data_ptr(), d_streams.data_ptr(), d_blocks.data_ptr(), len(frames),
                                                  out.data_ptr(), st.data_ptr(), current_stream(dev)))
        o, status = out.cpu().numpy(), st.cpu().numpy().tolist()
        at = 0
        for i, r in enumerate(raws):
            assert status[i] in (0, 1, 2, 3), status
            if frames[i] == good[i]:
                assert status[i] == 0 and o[at: at + len(r)].tobytes() == r
            assert not o[at + len(r): at + len(r) + gap].any(), (trial, i, "wrote behind its range")
            at += len(r) + gap


def test_lz4_frame_scanner_on_damaged_frames():
    """arx_lz4_frame_scan (a HOST function of the library) under random byte damage and truncation: it returns a status,
    its blocks never reach outside the frame it was given."""
    import ctypes as C

    from arrow_amd import _lib

    lib = _lib.get_lib()
    rng = np.random.default_rng(21)
    codec = pa.Codec("lz4")
    frames = [codec.compress(bytes(rng.integers(0, 7, 200_000, dtype=np.uint8))).to_pybytes(),
              codec.compress(bytes(rng.integers(0, 256, 150_000, dtype=np.uint8))).to_pybytes(), codec.compress(b"").to_pybytes()]
    for f in frames:
        nb = C.c_int64(0)
        assert lib.arx_lz4_frame_scan(f, len(f), 0, None, 0, C.byref(nb), None) == 0
        for trial in range(400):
            bad = bytearray(f)
            if trial % 3 == 0:
                bad = bad[: int(rng.integers(0, len(bad)))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, min(len(bad), 64 if trial % 2 else len(bad))))] = int(rng.integers(0, 256))
            bad = bytes(bad)
            tab = np.zeros(64, LZ4_BLOCK)
            nb = C.c_int64(0)
            rc = lib.arx_lz4_frame_scan(bad, len(bad), 0, tab.ctypes.data, 64, C.byref(nb), None)
            if rc == 0:
                assert 0 <= nb.value <= 64
                for b in tab[: nb.value]:
                    assert int(b["src_offset"]) + int(b["src_size"]) <= len(bad)


def test_ipc_record_batch_metadata_parser_against_pyarrow_and_garbage():
    """The RecordBatch flatbuffer parser of arrow_amd/ipc.py: nodes / buffer table / codec equal what pyarrow's reader
    reports for the same message; truncated or garbled metadata raises (struct.error / IndexError — the reader then
    falls back to pyarrow), never returns half a table."""
    import struct

    from arrow_amd import ipc as I

    rng = np.random.default_rng(9)
    t = _table(rng, 3000, 0.1)
    for compression in (None, "lz4", "zstd"):
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression=compression)) as w:
            w.write_table(t, max_chunksize=1000)
        msgs = [m for m in pa.ipc.MessageReader.open_stream(pa.BufferReader(sink.getvalue())) if m.type == "record batch"]
        assert len(msgs) == 3
        batches = list(pa.ipc.open_stream(pa.BufferReader(sink.getvalue())))
        for m, b in zip(msgs, batches):
            info = I.parse_record_batch_message(m.metadata.to_pybytes())
            assert info["length"] == b.num_rows
            assert info["codec"] == {None: None, "lz4": 0, "zstd": 1}[compression]
            assert [n for n, _ in info["nodes"]] == [b.num_rows] * b.num_columns
            assert [nc for _, nc in info["nodes"]] == [c.null_count for c in b.columns]
            assert len(info["buffers"]) == sum(3 if pa.types.is_string(f.type) else 2 for f in b.schema)
            assert all(off >= 0 and off + ln <= m.body.size for off, ln in info["buffers"])
        meta = msgs[0].metadata.to_pybytes()
        assert I.parse_record_batch_message(pa.ipc.MessageReader.open_stream(pa.BufferReader(sink.getvalue())).read_next_message()
                                            .metadata.to_pybytes()) is None      # the schema message is not a record batch
        for cut in (0, 3, 7, len(meta) // 2):
            with pytest.raises((struct.error, IndexError)):
                I.parse_record_batch_message(meta[:cut])
        for _ in range(300):                      # random single-byte damage: an exception or a table, never a hang / crash
            bad = bytearray(meta)
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
            try:
                I.parse_record_batch_message(bytes(bad))
            except (struct.error, IndexError, MemoryError, OverflowError):
                pass


@pytest.mark.emu
def test_ipc_corrupt_lz4_body(emu_ctx, tmp_path):
    _check_ipc_corrupt_lz4_body(emu_ctx, tmp_path)


@pytest.mark.gpu
def test_ipc_corrupt_lz4_body_gpu(gpu_ctx, tmp_path):
    _check_ipc_corrupt_lz4_body(gpu_ctx, tmp_path)


@pytest.mark.emu
def test_ipc_file_and_stream_to_device(emu_ctx, tmp_path):
    """IPC bodies are Arrow layout already: columns land on the device buffer by buffer and come back equal
    (fixed width, boolean, utf8, nulls, several batches, LZ4 body compression)."""
    _check_ipc(emu_ctx, tmp_path, 5000)


@pytest.mark.gpu
def test_ipc_file_and_stream_to_device_gpu(gpu_ctx, tmp_path):
    """The same on an MI355X: every buffer of every record batch is in HBM afterwards (device tensors), equal to the
    reference reader's arrays."""
    _check_ipc(gpu_ctx, tmp_path, 600_000)
    import torch

    rng = np.random.default_rng(5)
    t = _table(rng, 1000, 0.1)
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    for chunks in gpu_ctx.ipc.read_table(pa.BufferReader(sink.getvalue())).values():
        for arr in chunks:
            assert all(b is None or (isinstance(b, torch.Tensor) and b.is_cuda) for b in arr.buffers)


# ------------------------------------------------------------------ the delta / split encodings on the device
@pytest.mark.gpu
@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_byte_stream_split_gpu(gpu_ctx, tmp_path, null_p):
    _write_split_and_check(gpu_ctx, str(tmp_path), 600_000, null_p, 79, compression="snappy")


@pytest.mark.gpu
@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_delta_length_byte_array_gpu(gpu_ctx, tmp_path, null_p):
    _write_delta_length_and_check(gpu_ctx, str(tmp_path), 300_000, null_p, 83, compression="snappy")


@pytest.mark.gpu
def test_delta_decode_kernel_vs_restatement_gpu(gpu_ctx):
    PC.check_delta_decode(gpu_ctx, np.random.default_rng(5), 128, 4)
    PC.check_delta_decode(gpu_ctx, np.random.default_rng(6), 512, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_delta_binary_packed_gpu(gpu_ctx, tmp_path, null_p):
    _write_delta_and_check(gpu_ctx, str(tmp_path), 600_000, null_p, 77, compression="snappy")
    _write_delta_and_check(gpu_ctx, str(tmp_path), 70_001, null_p, 78, data_page_version="2.0", data_page_size=8192)


@pytest.mark.gpu
def test_delta_byte_array_kernels_vs_restatement_gpu(gpu_ctx):
    _check_dba_kernels(gpu_ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("null_p", [0.0, 0.1])
def test_parquet_delta_byte_array_gpu(gpu_ctx, tmp_path, null_p):
    _write_dba_and_check(gpu_ctx, str(tmp_path), 200_000, null_p, 93, compression="snappy")
    _write_dba_and_check(gpu_ctx, str(tmp_path), 30_001, null_p, 94, data_page_version="2.0", data_page_size=8192)


# --------------------------------------------------------------------------- repeated columns (lists)
# The reference's own known answers for DefRepLevelsToList / DefLevelsToBitmap:
# cpp/src/parquet/level_conversion_test.cc:142-161 (TriplyNestedList), :204-276 (outermost / middle / innermost list),
# :278-315 (SimpleLongList), :116-139 (WithRepetitionLevelFiltersOutEmptyListValues).
TRIPLY_DEF = [2, 7, 6, 7, 5, 3, 5, 5, 7, 7, 2, 7, 0, 1]
TRIPLY_REP = [0, 1, 3, 3, 2, 1, 0, 1, 2, 3, 1, 1, 0, 0]
LIST_VECTORS = [   # (def_level, rep_level, repeated_ancestor_def_level, offsets, validity, null_count)
    (2, 1, 0, [0, 3, 7, 7, 7], "1101", 1),
    (4, 2, 2, [0, 0, 2, 2, 3, 5, 5, 6], "0111101", 2),
    (6, 3, 4, [0, 3, 3, 3, 3, 5, 6], "111111", 0),
]


def test_oracle_list_levels_against_the_references_vectors():
    for dl, rl, anc, offsets, bits, nulls in LIST_VECTORS:
        got_off, got_valid, got_nulls = O.def_rep_levels_to_list(TRIPLY_DEF, TRIPLY_REP, dl, rl, anc)
        assert got_off.tolist() == offsets and "".join("1" if v else "0" for v in got_valid) == bits and got_nulls == nulls
    off, valid, nulls = O.def_rep_levels_to_list([2] * (65 * 9), ([0] + [1] * 8) * 65, 2, 1, 0)
    assert off.tolist() == [9 * x for x in range(66)] and valid.all() and len(valid) == 65 and nulls == 0
    got = O.def_levels_to_bitmap([0, 0, 0, 2, 2, 1, 0, 2], 2, 1, has_repeated_parent=True)
    assert "".join("1" if v else "0" for v in got) == "1101"       # (the test's bitmap "01101000" starts at bit 1)


def check_list_levels_kernel(amd, rng, scale=1):
    """arx_def_rep_levels_to_list / arx_levels_ge_bitmap against the oracle's slot-by-slot walk: the reference's vectors,
    then random level arrays (valid or not: the walk is defined for any input) of every size around the tile borders."""
    import ctypes as C

    import torch

    from arrow_amd import _lib
    from arrow_amd.array import alloc, to_device

    lib = _lib.get_lib()
    dev = amd.array.default_device()

    def run(d, r, dl, rl, anc, max_entries=None):
        n = len(d)
        d_def = to_device(np.asarray(d, np.uint32), dev)
        d_rep = to_device(np.asarray(r, np.uint32), dev)
        want_off, want_valid, want_nulls = O.def_rep_levels_to_list(d, r, dl, rl, anc)
        cap = len(want_valid) if max_entries is None else max_entries
        offsets = alloc((cap + 1) * 4, dev)
        valid = alloc((cap + 63) // 64 * 8 + 8, dev, zero=True)
        counts = torch.zeros(4, dtype=torch.int64, device=dev)
        ws = alloc(lib.arx_levels_to_list_workspace_bytes(n), dev)
        _lib.check(lib.arx_def_rep_levels_to_list(d_def.data_ptr(), d_rep.data_ptr(), n, dl, rl, anc, cap, offsets.data_ptr(),
                                                  valid.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), None))
        got = counts.cpu().numpy()
        assert got[0] == len(want_valid) and got[1] == want_off[-1], (got, len(want_valid), want_off[-1])
        if cap < len(want_valid):
            assert got[3] == 1
            return
        assert got[3] == 0 and got[2] == want_nulls
        assert np.array_equal(offsets.cpu().numpy().view(np.int32)[: cap + 1], want_off)
        bits = np.unpackbits(valid.cpu().numpy(), bitorder="little")
        assert np.array_equal(bits[:cap].astype(bool), want_valid) and not bits[cap:].any()
        for thr in (anc, dl):
            out = alloc((n + 63) // 64 * 8 + 8, dev)
            ones = torch.zeros(1, dtype=torch.int64, device=dev)
            _lib.check(lib.arx_levels_ge_bitmap(d_def.data_ptr(), n, thr, out.data_ptr(), ones.data_ptr(), None))
            want = np.asarray(d) >= thr
            have = np.unpackbits(out.cpu().numpy()[: (n + 63) // 64 * 8], bitorder="little")
            assert np.array_equal(have[:n].astype(bool), want) and not have[n:].any() and int(ones.item()) == want.sum()

    for dl, rl, anc, *_ in LIST_VECTORS:
        run(TRIPLY_DEF, TRIPLY_REP, dl, rl, anc)
    run([2] * (65 * 9), ([0] + [1] * 8) * 65, 2, 1, 0)
    run(TRIPLY_DEF, TRIPLY_REP, 2, 1, 0, max_entries=3)                    # "Definition levels exceeded upper bound"
    run([], [], 2, 1, 0)
    for n in [1, 63, 64, 255, 256, 257, 4095, 4096, 4097, 8192 + 5, 30_000 * scale]:
        for dl, rl, anc, max_def in [(2, 1, 0, 3), (1, 1, 0, 1), (4, 2, 2, 5), (6, 3, 4, 7)]:
            d = rng.integers(0, max_def + 1, n)
            r = rng.integers(0, rl + 2, n)
            r[0] = 0
            if n > 100:                # long lists and long stretches of skipped slots
                r[n // 3: n // 3 + n // 7] = rl
                d[n // 2: n // 2 + n // 9] = 0
            run(d, r, dl, rl, anc)


@pytest.mark.emu
def test_list_levels_kernel_emulator(emu_ctx):
    check_list_levels_kernel(emu_ctx, np.random.default_rng(12))


@pytest.mark.gpu
def test_list_levels_kernel_gpu(gpu_ctx):
    check_list_levels_kernel(gpu_ctx, np.random.default_rng(13), scale=40)


def _lists_of(rng, n, make_child, list_null_p, max_len, child_type):
    lens = rng.integers(0, max_len + 1, n)
    lens[rng.random(n) < 0.15] = 0                                     # empty lists next to null ones
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    child = make_child(int(offsets[-1]))
    mask = pa.array(rng.random(n) < list_null_p) if list_null_p else None
    return pa.ListArray.from_arrays(pa.array(offsets), child, type=child_type, mask=mask)


def _list_table(rng, n, null_p):
    def ints(k):
        return pa.array(rng.integers(-1000, 1000, k), mask=(rng.random(k) < null_p) if null_p else None)

    def strs(k):
        words = np.array(["", "a", "bb", "gfx950", "MI355X", "ünïcödé"], dtype=object)
        return pa.array(words[rng.integers(0, 6, k)], type=pa.string(), mask=(rng.random(k) < null_p) if null_p else None)

    def f64(k):
        return pa.array(np.round(rng.standard_normal(k), 1), mask=(rng.random(k) < null_p) if null_p else None)

    def flags(k):
        return pa.array(rng.random(k) < 0.4, type=pa.bool_(), mask=(rng.random(k) < null_p) if null_p else None)

    def i32_required(k):
        return pa.array(rng.integers(0, 50, k).astype(np.int32))

    def inner_lists(k):
        return _lists_of(rng, k, ints, null_p, 4, pa.list_(pa.field("element", pa.int64())))

    el = lambda t, nullable=True: pa.list_(pa.field("element", t, nullable=nullable))   # noqa: E731
    cols = {
        "l_i64": _lists_of(rng, n, ints, null_p, 6, el(pa.int64())),
        "l_str": _lists_of(rng, n, strs, null_p, 3, el(pa.string())),
        "l_f64": _lists_of(rng, n, f64, null_p, 5, el(pa.float64())),
        "l_flag": _lists_of(rng, n, flags, null_p, 9, el(pa.bool_())),
        "l_req": _lists_of(rng, n, i32_required, 0.0, 4, el(pa.int32(), nullable=False)),
        "ll_i64": _lists_of(rng, n, inner_lists, null_p, 3, el(el(pa.int64()))),
        "flat": pa.array(rng.integers(0, 9, n), mask=(rng.random(n) < null_p) if null_p else None),
    }
    fields = [pa.field(k, v.type, nullable=(k != "l_req")) for k, v in cols.items()]
    return pa.table(list(cols.values()), schema=pa.schema(fields))


LIST_VARIANTS = [dict(compression="snappy", data_page_version="1.0", use_dictionary=True),
                 dict(compression="none", data_page_version="2.0", use_dictionary=False, data_page_size=2048),
                 dict(compression="snappy", data_page_version="2.0", use_dictionary=["l_str", "l_i64"], data_page_size=4096),
                 dict(compression="zstd", data_page_version="1.0", use_dictionary=False, data_page_size=1024)]


def _write_and_check_lists(amd, tmp_path, n, null_p, variant, seed):
    rng = np.random.default_rng(seed)
    path = os.path.join(tmp_path, "lists.parquet")
    pq.write_table(_list_table(rng, n, null_p), path, row_group_size=max(1, n // 2 + 3), **variant)
    pf = pq.ParquetFile(path)
    names = [pf.metadata.schema.column(i).path for i in range(pf.metadata.num_columns)]
    got = amd.parquet.read_table(path)
    assert sorted(got) == sorted(names)
    for name, chunks in got.items():
        top = name.split(".")[0]
        for rg, arr in enumerate(chunks):
            want = pf.read_row_group(rg, columns=[top]).column(top).combine_chunks()
            have = arr.to_pyarrow()
            have.validate(full=True)
            assert have.type == want.type and len(have) == len(want) and have.null_count == want.null_count, (name, rg, have.type, want.type)
            assert have.equals(want), (name, rg, have.slice(0, 6), want.slice(0, 6))


@pytest.mark.emu
@pytest.mark.parametrize("variant", range(len(LIST_VARIANTS)))
@pytest.mark.parametrize("null_p", [0.0, 0.2])
def test_parquet_list_columns_emulator(emu_ctx, tmp_path, variant, null_p):
    """Repeated columns — list<T> and list<list<T>> of int64 / utf8 / float64 / bool / required int32 with null lists,
    empty lists and null elements, data pages V1 / V2 — equal to the reference reader's ListArrays."""
    _write_and_check_lists(emu_ctx, str(tmp_path), 3000, null_p, LIST_VARIANTS[variant], 300 + variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", range(len(LIST_VARIANTS)))
def test_parquet_list_columns_gpu(gpu_ctx, tmp_path, variant):
    _write_and_check_lists(gpu_ctx, str(tmp_path), 200_000, 0.1 if variant % 2 else 0.0, LIST_VARIANTS[variant], 400 + variant)


def test_parquet_struct_columns_are_refused_by_name(tmp_path):
    """What is NOT on the device path says so before any device call: structs (and lists of them)."""
    import arrow_amd

    path = os.path.join(str(tmp_path), "s.parquet")
    pq.write_table(pa.table({"s": pa.array([{"a": 1, "b": [1, 2]}, None])}), path)
    with pytest.raises(arrow_amd._lib.ArrowNotImplementedError, match="3-level LIST chain|not on the gfx950 path"):
        arrow_amd.parquet.read_table(path, columns=["s.b.list.element"], device="cpu")


# --------------------------------------------------------------------------- GZIP pages on the device
def check_gzip_kernel(amd, rng, scale=1):
    """arx_gzip_decompress_pages vs the reference codec's library (zlib, what GZipCodec wraps): gzip members and zlib streams
    (the codec auto-detects, compression_zlib.cc:88-95) at levels 0 (stored blocks), 1, 6, 9 and with the fixed Huffman code;
    empty, one byte, incompressible, long runs, periodic data whose matches overlap their own output, integers; a gzip header
    with every optional field; truncated / garbled / wrong-size blocks come back with a status and never write outside their
    destination."""
    import gzip
    import zlib

    import torch

    from arrow_amd import _lib
    from arrow_amd.array import current_stream, default_device, to_device

    lib, dev = _lib.get_lib(), default_device()

    def run(sizes, blocks):
        pages = np.zeros(len(blocks), SNAPPY_PAGE)
        so = do = 0
        for i, (n, b) in enumerate(zip(sizes, blocks)):
            pages[i] = (so, len(b), n, do)
            so += len(b)
            do += n
        src = to_device(np.frombuffer(b"".join(blocks) + b"\0" * 8, dtype=np.uint8), dev)
        out = torch.full((max(do, 1) + 64,), 0xEE, dtype=torch.uint8, device=dev)
        st = torch.full((len(blocks),), 77, dtype=torch.int32, device=dev)
        table = to_device(pages.view(np.uint8), dev)
        _lib.check(lib.arx_gzip_decompress_pages(src.data_ptr(), table.data_ptr(), len(blocks), out.data_ptr(),
                                                 st.data_ptr(), current_stream(dev)))
        host = out.cpu().numpy()
        return host[:do].tobytes(), st.cpu().numpy().tolist(), host[do:].tolist()

    raws = [b"", b"a", b"hello hello hello hello", bytes(rng.integers(0, 256, 5000 * scale, dtype=np.uint8)),
            bytes(np.repeat(rng.integers(0, 4, 300 * scale, dtype=np.uint8), 50)), np.cumsum(rng.integers(-3, 4, 20000 * scale)).tobytes(),
            b"x" * 100000 * scale, bytes(rng.integers(0, 3, 70000 * scale, dtype=np.uint8)), ("the quick brown fox " * 3000).encode(),
            np.round(rng.standard_normal(9000 * scale), 1).tobytes()]
    raws += [bytes(rng.integers(0, 256, period, dtype=np.uint8)) * (3000 // period + 2) for period in (1, 2, 3, 7, 63, 64, 65, 200, 257, 258, 259)]
    raws += [bytes(rng.integers(0, 256, 33_000, dtype=np.uint8)) * 2]          # matches at the far end of the 32 KB window
    blocks, sizes, want = [], [], []
    for r in raws:
        for level in (1, 6, 9, 0):
            blocks += [gzip.compress(r, level), zlib.compress(r, level)]
        fixed = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)        # the fixed Huffman code, gzip wrapper
        blocks.append(fixed.compress(r) + fixed.flush())
        several = zlib.compressobj(6, zlib.DEFLATED, 31)                       # several deflate blocks in one member
        blocks.append(several.compress(r[: len(r) // 2]) + several.flush(zlib.Z_FULL_FLUSH) + several.compress(r[len(r) // 2:]) + several.flush())
        sizes += [len(r)] * 10
        want += [r] * 10
    got, st, tail = run(sizes, blocks)
    assert st == [0] * len(blocks), [(i, x) for i, x in enumerate(st) if x]
    assert got == b"".join(want) and all(x == 0xEE for x in tail)
    # a gzip header with FEXTRA, FNAME, FCOMMENT and FHCRC (RFC 1952 2.3.1)
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = b"optional header fields " * 40
    deflated = body.compress(payload) + body.flush()
    import struct as _struct
    member = (bytes([0x1F, 0x8B, 8, 0x1E, 0, 0, 0, 0, 0, 3]) + _struct.pack("<H", 5) + b"extra" + b"name.bin\0" + b"a comment\0" + b"\x12\x34" +
              deflated + _struct.pack("<II", zlib.crc32(payload), len(payload)))
    got, st, _ = run([len(payload)], [member])
    assert st == [0] and got == payload
    # corrupt input: a status, and nothing outside the page's destination
    good = gzip.compress(raws[3], 6)
    garbled = bytearray(good)
    for k in range(len(good) // 3, len(good) // 3 + 40):
        garbled[k] ^= 0x5A
    cases = [(len(raws[3]), good[: len(good) // 2]),                 # truncated
             (len(raws[3]) + 1, good),                               # the page header announces another size (ISIZE check)
             (len(raws[3]), b"\x00\x01" + good[2:]),                 # neither a gzip nor a zlib header
             (len(raws[3]), bytes(garbled)),
             (10, bytes([0x78, 0x9C, 0x07]) + b"\0" * 8)]            # block type 3
    _, st, tail = run([c[0] for c in cases], [c[1] for c in cases])
    assert all(x != 0 for x in st[:3]) and st[4] == 4, st             # (garbage may decode to other bytes: then a size mismatch or not)
    assert all(x == 0xEE for x in tail)


@pytest.mark.emu
def test_gzip_page_decoder_kernel(emu_ctx):
    check_gzip_kernel(emu_ctx, np.random.default_rng(31))


@pytest.mark.gpu
def test_gzip_page_decoder_kernel_gpu(gpu_ctx):
    check_gzip_kernel(gpu_ctx, np.random.default_rng(32), scale=10)


def _gzip_plain_file(tmp_path, n, null_p, version):
    rng = np.random.default_rng(n + int(null_p * 100) + 1)
    mask = (rng.random(n) < null_p) if null_p else None
    t = pa.table({"opt": pa.array(np.cumsum(rng.integers(-3, 4, n)), mask=mask),
                  "req": pa.array(rng.integers(0, 50, n).astype(np.int32)),
                  "dbl": pa.array(np.round(rng.standard_normal(n), 1), mask=mask)})
    fields = [pa.field(f.name, f.type, nullable=(f.name != "req")) for f in t.schema]
    path = os.path.join(tmp_path, "gzip.parquet")
    pq.write_table(t.cast(pa.schema(fields)), path, use_dictionary=False, compression="gzip", data_page_version=version,
                   data_page_size=8192, row_group_size=n // 2 + 3)
    return path


@pytest.mark.emu
@pytest.mark.parametrize("version,null_p", [("2.0", 0.2), ("1.0", 0.0)])
def test_parquet_gzip_pages_decompressed_on_the_device(emu_ctx, tmp_path, version, null_p):
    """PLAIN fixed-width pages under GZIP take the Snappy pages' route (arx_gzip_decompress_pages); equal to pyarrow.parquet
    with the device route on and off."""
    path = _gzip_plain_file(str(tmp_path), 20000, null_p, version)
    stats = {}
    emu_ctx.parquet.read_table(path, stats=stats)
    assert stats.get("device_gzip_pages", 0) >= 2 * {"2.0": 3, "1.0": 1}[version], stats
    check_file(emu_ctx, path)
    emu_ctx.parquet.DEVICE_GZIP = False
    try:
        check_file(emu_ctx, path)
    finally:
        emu_ctx.parquet.DEVICE_GZIP = True


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["2.0", "1.0"])
def test_parquet_gzip_pages_decompressed_on_the_device_gpu(gpu_ctx, tmp_path, version):
    path = _gzip_plain_file(str(tmp_path), 600_000, 0.1, version)
    stats = {}
    gpu_ctx.parquet.read_table(path, stats=stats)
    assert stats.get("device_gzip_pages", 0) > 10, stats
    check_file(gpu_ctx, path)


# --------------------------------------------------------------------------- struct columns
def _struct_table(rng, n, struct_null_p, member_null_p):
    def m(p):
        return (rng.random(n) < p) if p else None

    typ = pa.struct([pa.field("a", pa.int64()), pa.field("b", pa.float64(), nullable=False), pa.field("s", pa.string()),
                     pa.field("f", pa.bool_()), pa.field("ts", pa.timestamp("us")), pa.field("i32", pa.int32())])
    members = [pa.array(rng.integers(0, 50, n), mask=m(member_null_p)), pa.array(np.round(rng.standard_normal(n), 2)),
               pa.array(np.array(["x", "yy", "", "gfx950"], dtype=object)[rng.integers(0, 4, n)], pa.string(), mask=m(member_null_p)),
               pa.array(rng.random(n) < 0.5, mask=m(member_null_p)), pa.array(rng.integers(0, 10**12, n), pa.timestamp("us"), mask=m(member_null_p)),
               pa.array(rng.integers(-9, 9, n).astype(np.int32), mask=m(member_null_p))]
    st = pa.StructArray.from_arrays(members, fields=list(typ), mask=pa.array(rng.random(n) < struct_null_p) if struct_null_p else None)
    req = pa.StructArray.from_arrays([pa.array(rng.integers(0, 9, n)), pa.array(rng.integers(0, 9, n), mask=m(member_null_p))],
                                     fields=[pa.field("x", pa.int64(), nullable=False), pa.field("y", pa.int64())])
    schema = pa.schema([pa.field("s", typ, nullable=bool(struct_null_p)), pa.field("flat", pa.int64()),
                        pa.field("r", req.type, nullable=False)])
    return pa.table([st, pa.array(np.arange(n)), req], schema=schema)


def _write_and_check_structs(amd, tmp_path, n, struct_null_p, member_null_p, variant, seed):
    rng = np.random.default_rng(seed)
    path = os.path.join(tmp_path, "structs.parquet")
    pq.write_table(_struct_table(rng, n, struct_null_p, member_null_p), path, row_group_size=max(1, n // 2 + 3), **variant)
    pf = pq.ParquetFile(path)
    got = amd.parquet.read_table(path)
    assert sorted(got) == ["flat", "r", "s"]
    assert sorted(amd.parquet.read_table(path, columns=["s"])) == ["s"]          # (nested columns by their top-level name)
    for name, chunks in got.items():
        assert len(chunks) == pf.metadata.num_row_groups
        for rg, arr in enumerate(chunks):
            want = pf.read_row_group(rg, columns=[name]).column(name).combine_chunks()
            have = arr.to_pyarrow()
            have.validate(full=True)
            assert have.type == want.type and have.null_count == want.null_count and have.equals(want), (name, rg, have.slice(0, 5), want.slice(0, 5))


@pytest.mark.emu
@pytest.mark.parametrize("variant", range(len(LIST_VARIANTS)))
@pytest.mark.parametrize("struct_null_p,member_null_p", [(0.15, 0.2), (0.0, 0.2), (0.3, 0.0)])
def test_parquet_struct_columns_emulator(emu_ctx, tmp_path, variant, struct_null_p, member_null_p):
    """Struct columns of primitives — nullable and required structs, nullable and required members (int64, float64, utf8,
    bool, timestamp, int32) — equal to the reference reader's StructArrays: members decoded as flat leaves with the struct's
    definition level on top, the struct's validity from a member's levels (arx_levels_ge_bitmap)."""
    v = dict(LIST_VARIANTS[variant])
    if isinstance(v.get("use_dictionary"), list):
        v["use_dictionary"] = True
    _write_and_check_structs(emu_ctx, str(tmp_path), 3000, struct_null_p, member_null_p, v, 500 + variant)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_parquet_struct_columns_gpu(gpu_ctx, tmp_path, variant):
    _write_and_check_structs(gpu_ctx, str(tmp_path), 200_000, 0.1, 0.15, dict(LIST_VARIANTS[variant]), 600 + variant)
