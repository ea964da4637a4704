"""Parquet column chunks decoded into HBM (first slice).

Scope: flat columns of physical type INT32 / INT64 / FLOAT / DOUBLE (PLAIN and PLAIN_DICTIONARY /
RLE_DICTIONARY encodings, incl. the dictionary -> PLAIN fallback inside a chunk; DELTA_BINARY_PACKED for
INT32 / INT64; BYTE_STREAM_SPLIT), BOOLEAN (PLAIN, RLE) and
BYTE_ARRAY columns (utf8 / binary; dictionary-encoded, PLAIN, DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY pages); required or optional (max definition level <= 1, no repetition),
data pages V1 and V2, any page compression pyarrow's codecs can undo.

Division of labour (what the reference does in cpp/src/parquet/column_reader.cc:740-1000 and
decoder.cc on the CPU):
  host  : file metadata (pyarrow.parquet metadata API), page headers (Thrift compact protocol,
          cpp/src/parquet/parquet.thrift:811-845), page decompression, and ONE walk over the
          variable-length run headers of the RLE / bit-packed hybrids (a few bytes per run);
  device: definition levels -> validity bitmap, dictionary indices -> values (arx_rle_decode_*,
          arx_take), and the spreading of the non-null values over their slots
          (arx_filter_count + arx_expand_by_mask).  Encoded bytes cross PCIe, decoded columns never do.
One device Array per row group (what pyarrow returns as the chunks of a ChunkedArray)."""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np
import torch

from . import _lib
from ._lib import ArrowInvalid, ArrowNotImplementedError, check
from .array import Array, alloc, bitmap_nbytes, current_stream, default_device, float32, float64, int32, int64, to_device

# parquet.thrift enums
_PAGE_DATA, _PAGE_INDEX, _PAGE_DICT, _PAGE_DATA_V2 = 0, 1, 2, 3
_ENC_PLAIN, _ENC_PLAIN_DICT, _ENC_RLE, _ENC_BIT_PACKED, _ENC_RLE_DICT = 0, 2, 3, 4, 8
_ENC_DELTA_BINARY_PACKED, _ENC_DELTA_LENGTH_BYTE_ARRAY, _ENC_DELTA_BYTE_ARRAY, _ENC_BYTE_STREAM_SPLIT = 5, 6, 7, 9
_PHYSICAL = {"INT32": (int32, np.int32), "INT64": (int64, np.int64), "FLOAT": (float32, np.float32),
             "DOUBLE": (float64, np.float64)}

RUN_DTYPE = np.dtype([("out_start", "<u4"), ("kind", "<u4"), ("payload", "<u8")])   # struct ArxRleRun
MINIBLOCK_DTYPE = np.dtype([("bit_start", "<u8"), ("min_delta", "<i8"), ("bit_width", "<u4"), ("reserved", "<u4")])   # struct ArxDeltaMiniblock
DELTA_PAGE_DTYPE = np.dtype([("out_start", "<i8"), ("first_miniblock", "<i8"), ("values_per_miniblock", "<i8"),
                             ("first_value", "<i8"), ("num_values", "<i8"), ("first_tile", "<i8")])   # ArxDeltaPage


# --------------------------------------------------------------------------- Thrift compact protocol
class _Thrift:
    """Just enough of the compact protocol (thrift/protocol/TCompactProtocol) to read a PageHeader."""

    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def varint(self):
        r, s = 0, 0
        while True:
            c = self.b[self.p]
            self.p += 1
            r |= (c & 0x7F) << s
            if not c & 0x80:
                return r
            s += 7

    def zigzag(self):
        v = self.varint()
        return (v >> 1) ^ -(v & 1)

    def skip(self, t):
        if t in (1, 2):
            return
        if t == 3:
            self.p += 1
        elif t in (4, 5, 6):
            self.varint()
        elif t == 7:
            self.p += 8
        elif t == 8:
            n = self.varint()      # (not `self.p += self.varint()`: the left side would be read first)
            self.p += n
        elif t in (9, 10):
            h = self.b[self.p]
            self.p += 1
            n = h >> 4
            if n == 15:
                n = self.varint()
            for _ in range(n):
                self.skip(h & 0x0F)
        elif t == 11:
            n = self.varint()
            if n:
                kv = self.b[self.p]
                self.p += 1
                for _ in range(n):
                    self.skip(kv >> 4)
                    self.skip(kv & 0x0F)
        elif t == 12:
            self.struct(None)
        else:
            raise ArrowInvalid(f"Parquet page header: unknown Thrift type {t}")

    def struct(self, wanted):
        """Reads one struct; `wanted` maps field id -> 'i' (integer), 'b' (bool) or a nested map."""
        out, fid = {}, 0
        while True:
            h = self.b[self.p]
            self.p += 1
            if h == 0:
                return out
            t = h & 0x0F
            fid = fid + (h >> 4) if h >> 4 else self.zigzag()
            kind = None if wanted is None else wanted.get(fid)
            if kind == "i" and t in (4, 5, 6):
                out[fid] = self.zigzag()
            elif kind == "b" and t in (1, 2):
                out[fid] = t == 1
            elif isinstance(kind, dict) and t == 12:
                out[fid] = self.struct(kind)
            else:
                self.skip(t)


_PAGE_HEADER = {1: "i", 2: "i", 3: "i",
                5: {1: "i", 2: "i", 3: "i", 4: "i"},                 # DataPageHeader
                7: {1: "i", 2: "i"},                                 # DictionaryPageHeader
                8: {1: "i", 2: "i", 3: "i", 4: "i", 5: "i", 6: "i", 7: "b"}}   # DataPageHeaderV2


def read_page_header(buf, pos):
    """(header dict, position of the page payload)."""
    t = _Thrift(buf, pos)
    return t.struct(_PAGE_HEADER), t.p


# --------------------------------------------------------------------------- run headers (host walk)
def scan_delta_miniblocks(data, byte_base: int = 0):
    """Walks the block headers of one DELTA_BINARY_PACKED page (DeltaBitPackDecoder::InitHeader / InitBlock,
    parquet/decoder.cc) with the host function arx_delta_scan_miniblocks.  Returns (miniblocks, values per
    miniblock, total values, first value, bytes consumed); bit positions are shifted by byte_base."""
    lib = _lib.get_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    ptr = buf.ctypes.data if len(buf) else None
    nmb, vpm, total, first, used = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_size_t(0)
    check(lib.arx_delta_scan_miniblocks(ptr, len(buf), byte_base, None, 0, C.byref(nmb), C.byref(vpm), C.byref(total),
                                        C.byref(first), C.byref(used)))           # pass 1: count
    mbs = np.zeros(max(nmb.value, 1), dtype=MINIBLOCK_DTYPE)
    check(lib.arx_delta_scan_miniblocks(ptr, len(buf), byte_base, mbs.ctypes.data, nmb.value, C.byref(nmb), C.byref(vpm),
                                        C.byref(total), C.byref(first), C.byref(used)))     # pass 2: fill
    return mbs[: nmb.value], vpm.value, total.value, first.value, used.value


def decode_delta_binary_packed(data, byte_width: int = 8, device=None) -> Array:
    """One DELTA_BINARY_PACKED block sequence (a data page's values) -> an int32 / int64 device array:
    header walk on the host, unpack + prefix sum on the device (arx_delta_decode)."""
    device = torch.device(device) if device is not None else default_device()
    lib, stream = _lib.get_lib(), current_stream(device)
    mbs, vpm, total, first, used = scan_delta_miniblocks(data)
    atype, _ = _PHYSICAL["INT64" if byte_width == 8 else "INT32"]
    out = alloc(total * byte_width, device)
    d_bytes = to_device(np.frombuffer(bytes(data[:used]) + b"\0" * (16 + (-used % 8)), dtype=np.uint8), device)
    table = mbs if len(mbs) else np.zeros(1, MINIBLOCK_DTYPE)
    d_table = to_device(table.view(np.uint8), device)
    ws = alloc(lib.arx_delta_decode_workspace_bytes(total), device)
    check(lib.arx_delta_decode(d_bytes.data_ptr(), d_table.data_ptr(), len(mbs), vpm, first, total, byte_width,
                               ws.data_ptr(), ws.numel(), out.data_ptr(), stream))
    return Array(atype, total, [None, out], 0, 0)


def decode_delta_streams(streams, byte_width: int, device=None):
    """Several DELTA_BINARY_PACKED streams (the head of each byte string) decoded into ONE device buffer, stream after
    stream, by one launch sequence (arx_delta_decode_pages: every stream owns whole 4096-value tiles).  Returns (the uint8
    tensor of all values, [(values in the stream, bytes it takes)])."""
    device = torch.device(device) if device is not None else default_device()
    lib, stream = _lib.get_lib(), current_stream(device)
    delta_bytes, tables, info = bytearray(), [], []
    pages = np.zeros(max(len(streams), 1), dtype=DELTA_PAGE_DTYPE)
    start = at = tiles = 0
    for j, data in enumerate(streams):
        mbs, vpm, total, first, used = scan_delta_miniblocks(data, byte_base=len(delta_bytes))
        pages[j] = (start, at, max(vpm, 1), first, total, tiles)
        tables.append(mbs)
        delta_bytes += bytes(data[:used])
        delta_bytes += b"\0" * (-len(delta_bytes) % 8)      # the next stream starts on a 64-bit word
        info.append((total, used))
        start += total
        at += len(mbs)
        tiles += (total + 4095) // 4096
    out = alloc(max(start, 1) * byte_width, device)
    if start:
        d_bytes = to_device(np.frombuffer(bytes(delta_bytes) + b"\0" * 16, dtype=np.uint8), device)
        table = np.concatenate(tables) if at else np.zeros(1, MINIBLOCK_DTYPE)
        d_table = to_device(table.view(np.uint8), device)
        d_pages = to_device(pages.view(np.uint8), device)
        ws = alloc(lib.arx_delta_decode_workspace_bytes(max(tiles, 1) * 4096), device)
        check(lib.arx_delta_decode_pages(d_bytes.data_ptr(), d_table.data_ptr(), d_pages.data_ptr(), len(streams), tiles, byte_width,
                                         ws.data_ptr(), ws.numel(), out.data_ptr(), stream))
    return out, info


_DBA_STATUS = ((1, "negative prefix length in DELTA_BYTE_ARRAY"), (2, "prefix length too large in DELTA_BYTE_ARRAY"),
               (4, "negative suffix length in DELTA_BYTE_ARRAY"), (8, "excess expansion in DELTA_BYTE_ARRAY"),
               (16, "DELTA_BYTE_ARRAY suffix lengths do not add up to the bytes their page carries"))


def expand_delta_byte_array(pages, base: int = 0, offsets_out: torch.Tensor | None = None, device=None):
    """DELTA_BYTE_ARRAY data pages (DeltaByteArrayDecoderImpl, parquet/decoder.cc:1974-2204) -> offsets + bytes in HBM.
    `pages`: [(value bytes of the page, number of values)], one column chunk's pages in order.  Host: the two
    DELTA_BINARY_PACKED header walks per page (prefix lengths; the suffix block's lengths).  Device: both length
    streams (arx_delta_decode), value lengths + the decoder's checks (arx_delta_byte_array_lengths), the two offset
    scans (arx_lengths_to_offsets_i32) and the expansion, one wave per page (arx_delta_byte_array_expand).
    Returns (offsets int32 tensor view of count + 1 entries starting at `base` — written into `offsets_out` when given —,
    the expanded bytes as a uint8 tensor, count)."""
    device = torch.device(device) if device is not None else default_device()
    lib, stream = _lib.get_lib(), current_stream(device)
    pages = [(bytes(page), count) for page, count in pages if count]
    # both length streams of every page in ONE launch sequence: [prefix lengths of page 0, 1, ... | suffix lengths of page 0, 1, ...]
    heads = [page for page, _ in pages]
    rests = []
    for page in heads:                      # the suffix block starts where the prefix stream ends
        _, _, _, _, used = scan_delta_miniblocks(page)
        rests.append(page[used:])
    lens, info = decode_delta_streams(heads + rests, 4, device)
    npages = len(pages)
    firsts, suffix_starts, suffix_parts = [0], [0], []
    for (page, count), (ptotal, _), (stotal, sused), rest in zip(pages, info[:npages], info[npages:], rests):
        if ptotal != count:
            raise ArrowInvalid(f"Parquet: DELTA_BYTE_ARRAY page holds {ptotal} prefix lengths, its header says {count}")
        if stotal != count:
            raise ArrowInvalid(f"Parquet: DELTA_BYTE_ARRAY page holds {stotal} suffix lengths and {count} prefix lengths")
        suffix_parts.append(rest[sused:])
        firsts.append(firsts[-1] + count)
        suffix_starts.append(suffix_starts[-1] + len(rest) - sused)
    n = firsts[-1]
    if offsets_out is None:
        offsets_out = alloc((n + 1) * 4, device)
    if n == 0:
        offsets_out.view(torch.int32)[0] = base
        return offsets_out.view(torch.int32)[:1], alloc(0, device), 0
    if suffix_starts[-1] > 2**31 - 1:
        raise ArrowInvalid("Parquet: column chunk exceeds the int32 offset range")
    prefix, slen = lens[: n * 4], lens[n * 4: 2 * n * 4]
    d_first = to_device(np.asarray(firsts, dtype=np.int64).view(np.uint8), device)
    out_len = alloc(n * 4, device)
    state = alloc(16, device)      # [status bits, bytes of all values]
    check(lib.arx_delta_byte_array_lengths(prefix.data_ptr(), slen.data_ptr(), n, d_first.data_ptr(), npages, out_len.data_ptr(),
                                           state.data_ptr(), stream))
    ws = alloc(lib.arx_delta_decode_workspace_bytes(n + 1), device)
    soff = alloc((n + 1) * 4, device)
    check(lib.arx_lengths_to_offsets_i32(slen.data_ptr(), n, 0, soff.data_ptr(), ws.data_ptr(), ws.numel(), stream))
    check(lib.arx_lengths_to_offsets_i32(out_len.data_ptr(), n, base, offsets_out.data_ptr(), ws.data_ptr(), ws.numel(), stream))
    flags, total = (int(x) for x in state.view(torch.int64)[:2].cpu().numpy())
    for bit, text in _DBA_STATUS:      # corrupt lengths stop here: the expansion only ever sees lengths that passed the checks
        if flags & bit:
            raise ArrowInvalid("Parquet: " + text)
    if base + total > 2**31 - 1:
        raise ArrowInvalid("Parquet: column chunk exceeds the int32 offset range")
    end = base + total
    suffix = b"".join(suffix_parts)
    d_suffix = to_device(np.frombuffer(suffix + b"\0" * (8 + (-len(suffix) % 8)), dtype=np.uint8), device)
    d_sfirst = to_device(np.asarray(suffix_starts, dtype=np.int64).view(np.uint8), device)
    out = alloc(max(end - base, 1), device)
    check(lib.arx_delta_byte_array_expand(prefix.data_ptr(), soff.data_ptr(), d_suffix.data_ptr(), len(suffix), offsets_out.data_ptr(),
                                          base, d_first.data_ptr(), d_sfirst.data_ptr(), npages, out.data_ptr(), state.data_ptr(),
                                          stream))
    bad = int(state.view(torch.int64)[0].item())
    for bit, text in _DBA_STATUS:
        if bad & bit:
            raise ArrowInvalid("Parquet: " + text)
    return offsets_out.view(torch.int32)[: n + 1], out[: end - base], n


def decode_delta_byte_array(pages, atype=None, device=None) -> Array:
    """The values of DELTA_BYTE_ARRAY pages as one binary (or `atype`) device array — see expand_delta_byte_array."""
    from .array import binary

    offs, data, n = expand_delta_byte_array(pages, 0, None, device)
    if data.numel() == 0:
        data = alloc(1, offs.device)
    return Array(atype or binary, n, [None, offs.view(torch.uint8), data], 0, 0)


def scan_rle_runs(data, bit_width: int, num_values: int, out_base: int = 0, byte_base: int = 0, equals: int | None = None):
    """Walks the run headers of an RLE / bit-packed hybrid block (rle_encoding_internal.h:40-90) until
    `num_values` values are covered — the host function arx_rle_scan_runs of the library (the walk is
    sequential and touches a few bytes per run).  Returns (runs, ones): a RUN_DTYPE array whose
    out_start / payload are shifted by out_base / byte_base (so the pages of a chunk share one table)
    and, for bit_width == 1, the number of values equal to 1 (the non-null count of a level block).  `equals`
    (levels of a nested column, any width up to 16 bits): count the values equal to it instead
    (arx_rle_scan_runs_equals)."""
    lib = _lib.get_lib()
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    nbytes = len(buf)
    ptr = buf.ctypes.data if nbytes else None
    nr, ones = C.c_int64(0), C.c_int64(0)
    if equals is not None:
        check(lib.arx_rle_scan_runs_equals(ptr, nbytes, bit_width, num_values, equals, out_base, byte_base, None, 0,
                                           C.byref(nr), C.byref(ones)))
        runs = np.zeros(max(nr.value, 1), dtype=RUN_DTYPE)
        check(lib.arx_rle_scan_runs_equals(ptr, nbytes, bit_width, num_values, equals, out_base, byte_base,
                                           runs.ctypes.data, nr.value, C.byref(nr), None))
        return runs[: nr.value], ones.value
    check(lib.arx_rle_scan_runs(ptr, nbytes, bit_width, num_values, out_base, byte_base, None, 0,
                                C.byref(nr), C.byref(ones)))                      # pass 1: count
    runs = np.zeros(max(nr.value, 1), dtype=RUN_DTYPE)
    check(lib.arx_rle_scan_runs(ptr, nbytes, bit_width, num_values, out_base, byte_base, runs.ctypes.data, nr.value,
                                C.byref(nr), None))                               # pass 2: fill
    return runs[: nr.value], ones.value


# Snappy pages whose uncompressed bytes the host never needs (PLAIN fixed-width values: data page V2 bodies — their
# levels travel uncompressed — and V1 pages of required columns) are decompressed in HBM (arx_snappy_decompress_pages):
# the COMPRESSED bytes cross PCIe and no host codec runs.  Pages with run headers to walk (levels inside a V1 block,
# dictionary indices, delta blocks) still go through the host codec: the walk needs their bytes.
DEVICE_SNAPPY = True
# GZIP pages of the same kinds go the same way (arx_gzip_decompress_pages: RFC 1952 / 1950 / 1951 on the device).
DEVICE_GZIP = True
SNAPPY_PAGE_DTYPE = np.dtype([("src_offset", "<u8"), ("src_size", "<u4"), ("dst_size", "<u4"), ("dst_offset", "<u8")])  # struct ArxSnappyPage


# --------------------------------------------------------------------------- one column chunk
def _decompress(codec: str, payload, uncompressed_size: int):
    if codec == "UNCOMPRESSED":
        return bytes(payload)
    import pyarrow as pa

    if codec == "LZ4":     # the legacy Hadoop-framed LZ4 of Parquet; pyarrow exposes no stand-alone codec for it
        raise ArrowNotImplementedError("Parquet: legacy LZ4 (Hadoop framing) pages; LZ4_RAW is supported")
    return pa.Codec(codec.lower()).decompress(bytes(payload), decompressed_size=uncompressed_size).to_pybytes()


def _column_chunk_pages(raw, col):
    """Yields (header, decompressed payload bytes) for every page of a column chunk; `raw` = the file bytes."""
    start = col.data_page_offset
    if col.has_dictionary_page and col.dictionary_page_offset is not None:
        start = min(start, col.dictionary_page_offset)
    pos, seen = start, 0
    while seen < col.num_values:          # (SerializedPageReader::NextPage stops on the value count as well)
        hdr, body = read_page_header(raw, pos)
        csize = hdr[3]
        payload = raw[body: body + csize]
        pos = body + csize
        if hdr[1] == _PAGE_DATA:
            seen += hdr[5][1]
        elif hdr[1] == _PAGE_DATA_V2:
            seen += hdr[8][1]
        yield hdr, payload


def _device_runs(runs: np.ndarray, device):
    return to_device(runs.view(np.uint8) if len(runs) else np.zeros(16, np.uint8), device)


def _parse_byte_array_dictionary(page: bytes, count: int):
    """PLAIN BYTE_ARRAY values (4-byte little-endian length + bytes each, PlainByteArrayDecoder in
    parquet/decoder.cc) -> (int32 offsets[count + 1], data bytes).  Dictionary pages only: small."""
    offsets = np.zeros(count + 1, dtype=np.int32)
    parts, pos, total = [], 0, 0
    for i in range(count):
        (ln,) = struct.unpack_from("<i", page, pos)
        if ln < 0 or pos + 4 + ln > len(page):
            raise ArrowInvalid("Parquet: byte-array dictionary entry runs past the page (corrupt page?)")
        parts.append(page[pos + 4: pos + 4 + ln])
        pos += 4 + ln
        total += ln
        offsets[i + 1] = total
    return offsets, b"".join(parts)


def read_column_chunk(raw, col, max_def_level: int, device=None, stats: dict | None = None,
                      binary_type=None, nested: dict | None = None) -> Array:
    """Decodes one column chunk (all its pages) into a device Array.  `stats` (optional) accumulates
    host_prep_s: seconds spent on the host before the first device call (headers, decompression, run walk).
    `nested` (a repeated column; {"max_rep": its maximum repetition level}): the Array that comes back has one slot per
    LEVEL (valid where def == max_def_level: the values spread over all level slots), and the dict receives the level
    blocks and run tables of the chunk ("def_bytes", "def_runs", "rep_bytes", "rep_runs", "levels") for
    assemble_list_column."""
    import time

    t_start = time.perf_counter()
    device = torch.device(device) if device is not None else default_device()
    is_binary = col.physical_type == "BYTE_ARRAY" and binary_type is not None
    is_bool = col.physical_type == "BOOLEAN"
    if col.physical_type not in _PHYSICAL and not is_binary and not is_bool:
        raise ArrowNotImplementedError(f"Parquet physical type {col.physical_type} is not on the gfx950 path")
    if max_def_level > 1 and nested is None:
        raise ArrowNotImplementedError("Parquet: nested / repeated columns are not on the gfx950 path")
    max_rep_level = nested["max_rep"] if nested is not None else 0
    def_bw, rep_bw = max(1, int(max_def_level).bit_length()), int(max_rep_level).bit_length()
    def_equals = max_def_level if nested is not None else None     # (flat columns keep the bit-width-1 popcount walk)
    rep_bytes, rep_runs = bytearray(), []
    if is_binary:
        atype, width = binary_type, 1          # values live in the dictionary; data pages carry indices only
    elif is_bool:
        from .array import bool_

        atype, width = bool_, 1                # bit-packed values (PLAIN) or the bit-width-1 hybrid (RLE)
    else:
        atype, np_dtype = _PHYSICAL[col.physical_type]
        width = np.dtype(np_dtype).itemsize
    lib, stream = _lib.get_lib(), current_stream(device)
    codec = col.compression

    dict_bytes, dict_count = None, 0
    level_bytes, level_runs = bytearray(), []
    index_bytes, index_runs = bytearray(), []
    plain_segments = []       # host-decoded PLAIN pages: (byte position among the PLAIN values, bytes)
    plain_pos = 0             # bytes of PLAIN values so far (host- and device-decoded pages alike)
    device_snappy_pages = []  # (compressed block, uncompressed size, byte position among the PLAIN values)
    device_codec = (DEVICE_SNAPPY and codec == "SNAPPY") or (DEVICE_GZIP and codec == "GZIP")   # pages decompressed in HBM
    plain_pages = []          # BYTE_ARRAY only: (page value bytes, number of values)
    bool_bytes, bool_runs = bytearray(), []   # BOOLEAN only: every page becomes runs of one shared table
    split_pages = []                            # BYTE_STREAM_SPLIT: (first dense slot, page bytes, count)
    delta_bytes, delta_pages = bytearray(), []  # DELTA_BINARY_PACKED: (first dense slot, miniblocks, per miniblock, first value, count)
    rows, dense, dense_from_dict = 0, 0, 0
    for hdr, payload in _column_chunk_pages(raw, col):
        ptype = hdr[1]
        if ptype == _PAGE_DICT:
            dh = hdr[7]
            if dh.get(2, _ENC_PLAIN) not in (_ENC_PLAIN, _ENC_PLAIN_DICT):
                raise ArrowNotImplementedError("Parquet: dictionary page encoding")
            dict_bytes, dict_count = _decompress(codec, payload, hdr[2]), dh[1]
            continue
        if ptype == _PAGE_DATA:
            dh = hdr[5]
            nvals, enc = dh[1], dh[2]
            on_device = (device_codec and enc == _ENC_PLAIN and max_def_level == 0 and
                         max_rep_level == 0 and not is_binary and not is_bool and hdr[2] == nvals * width)
            if on_device:
                device_snappy_pages.append((bytes(payload), hdr[2], plain_pos))
                plain_pos += hdr[2]                   # (the device writes these bytes: nothing is staged for them)
                rows += nvals
                dense += nvals
                continue
            page = _decompress(codec, payload, hdr[2])
            pos = 0
            if max_rep_level > 0:                                  # repetition levels come first (column_reader.cc:784-)
                if dh[4] != _ENC_RLE:
                    raise ArrowNotImplementedError("Parquet: BIT_PACKED repetition levels")
                (nbytes,) = struct.unpack_from("<i", page, 0)
                rlevels = page[4: 4 + nbytes]
                pos = 4 + nbytes
            if max_def_level > 0:
                if dh[3] != _ENC_RLE:
                    raise ArrowNotImplementedError("Parquet: BIT_PACKED definition levels")
                (nbytes,) = struct.unpack_from("<i", page, pos)    # LevelDecoder::SetData, column_reader.cc:139
                levels = page[pos + 4: pos + 4 + nbytes]
                pos += 4 + nbytes
        elif ptype == _PAGE_DATA_V2:
            dh = hdr[8]
            nvals, enc = dh[1], dh[4]
            dl, rl = dh.get(5, 0), dh.get(6, 0)
            if rl and max_rep_level == 0:
                raise ArrowNotImplementedError("Parquet: repetition levels")
            rlevels = bytes(payload[:rl])                             # V2 levels are never compressed: repetition,
            levels = bytes(payload[rl: rl + dl])                      # then definition levels, then the values
            body = payload[rl + dl:]
            on_device = (device_codec and dh.get(7, True) and enc == _ENC_PLAIN and
                         not is_binary and not is_bool)
            if on_device:
                valid_here = nvals
                if max_def_level > 0:
                    runs, ones = scan_rle_runs(levels, def_bw, nvals, out_base=rows, byte_base=len(level_bytes),
                                               equals=def_equals)
                    valid_here = ones
                if hdr[2] - dl - rl == valid_here * width:
                    if max_def_level > 0:
                        level_runs.append(runs)
                        level_bytes += levels
                    if max_rep_level > 0:
                        rep_runs.append(scan_rle_runs(rlevels, rep_bw, nvals, out_base=rows, byte_base=len(rep_bytes))[0])
                        rep_bytes += rlevels
                    device_snappy_pages.append((bytes(body), hdr[2] - dl - rl, plain_pos))
                    plain_pos += hdr[2] - dl - rl
                    rows += nvals
                    dense += valid_here
                    continue
            page = _decompress(codec, body, hdr[2] - dl - rl) if dh.get(7, True) else bytes(body)
            pos = 0
        else:
            continue                                                  # index pages etc.
        valid_here = nvals
        if max_def_level > 0:
            runs, ones = scan_rle_runs(levels, def_bw, nvals, out_base=rows, byte_base=len(level_bytes), equals=def_equals)
            level_runs.append(runs)
            level_bytes += levels
            valid_here = ones
        if max_rep_level > 0:
            rep_runs.append(scan_rle_runs(rlevels, rep_bw, nvals, out_base=rows, byte_base=len(rep_bytes))[0])
            rep_bytes += rlevels
        values = page[pos:]
        if enc in (_ENC_PLAIN_DICT, _ENC_RLE_DICT):
            if plain_pos or plain_pages:
                raise ArrowNotImplementedError("Parquet: a dictionary-encoded page after a PLAIN page in one column chunk")
            bw = values[0] if len(values) else 0
            if valid_here:
                runs, _ = scan_rle_runs(values[1:], bw, valid_here, out_base=dense, byte_base=len(index_bytes))
                # every page may use its own bit width (a growing dictionary): kept per run in kind >> 8
                runs["kind"] |= np.uint32(bw << 8)
                index_runs.append(runs)
                index_bytes += values[1:]
            dense_from_dict += valid_here
        elif is_bool and enc in (_ENC_PLAIN, _ENC_RLE):
            if valid_here:
                if enc == _ENC_PLAIN:      # LSB-first bit-packed values = one literal run of the hybrid's bit layout
                    run = np.zeros(1, dtype=RUN_DTYPE)
                    run["out_start"], run["kind"], run["payload"] = dense, 1, len(bool_bytes)
                    bool_runs.append(run)
                    bool_bytes += values[: (valid_here + 7) // 8]
                else:                      # RLE: 4-byte length + hybrid, bit width 1 (parquet/decoder.cc, RleBooleanDecoder)
                    (nb,) = struct.unpack_from("<i", values, 0)
                    runs, _ = scan_rle_runs(values[4: 4 + nb], 1, valid_here, out_base=dense, byte_base=len(bool_bytes))
                    bool_runs.append(runs)
                    bool_bytes += values[4: 4 + nb]
        elif enc == _ENC_PLAIN:
            if is_binary:
                plain_pages.append((bytes(values), valid_here, "plain"))      # offsets are built once the dictionary size is known
            else:
                plain_segments.append((plain_pos, bytes(values[: valid_here * width])))
                plain_pos += valid_here * width
        elif enc == _ENC_DELTA_LENGTH_BYTE_ARRAY and is_binary:
            plain_pages.append((bytes(values), valid_here, "delta_length"))
        elif enc == _ENC_DELTA_BYTE_ARRAY and is_binary:
            plain_pages.append((bytes(values), valid_here, "delta_byte_array"))
        elif enc == _ENC_BYTE_STREAM_SPLIT and not is_binary and not is_bool:
            if len(values) < valid_here * width:
                raise ArrowInvalid("Parquet: BYTE_STREAM_SPLIT page shorter than its values")
            if valid_here:
                split_pages.append((dense, bytes(values[: valid_here * width]), valid_here))
        elif enc == _ENC_DELTA_BINARY_PACKED and col.physical_type in ("INT32", "INT64"):
            if valid_here:
                mbs, vpm, total, first, used = scan_delta_miniblocks(values, byte_base=len(delta_bytes))
                if total != valid_here:
                    raise ArrowInvalid(f"Parquet: DELTA_BINARY_PACKED page holds {total} values, its header says {valid_here}")
                delta_pages.append((dense, mbs, vpm, first, valid_here))
                delta_bytes += values[:used]
                delta_bytes += b"\0" * (-len(delta_bytes) % 8)      # the next page starts on a 64-bit word
        else:
            raise ArrowNotImplementedError(f"Parquet encoding {enc} is not on the gfx950 path")
        rows += nvals
        dense += valid_here

    if nested is not None:
        nested.update(def_bytes=bytes(level_bytes), def_runs=list(level_runs), rep_bytes=bytes(rep_bytes),
                      rep_runs=rep_runs, levels=rows, def_bw=def_bw, rep_bw=rep_bw)
    if stats is not None:
        stats["host_prep_s"] = stats.get("host_prep_s", 0.0) + (time.perf_counter() - t_start)
        stats["encoded_bytes"] = stats.get("encoded_bytes", 0) + len(level_bytes) + len(index_bytes) + sum(len(b) for _, b in plain_segments) + len(dict_bytes or b"")
    if is_bool:
        return _finish_boolean_chunk(lib, stream, device, bool_bytes, bool_runs, level_bytes, level_runs, rows, dense,
                                     max_def_level)
    if is_binary:
        return _finish_binary_chunk(lib, stream, device, atype, dict_bytes, dict_count, index_bytes, index_runs,
                                    level_bytes, level_runs, rows, dense, max_def_level, dense_from_dict, plain_pages)
    # ---- dense values in HBM.  A chunk may start dictionary-encoded and fall back to PLAIN once the
    # dictionary outgrows its page (ColumnWriterImpl::FallbackToPlainEncoding, parquet/column_writer.cc):
    # the dense buffer is then [values of the dictionary pages][values of the PLAIN pages].
    dense_buf = alloc(dense * width, device)
    if index_runs:
        runs = np.concatenate(index_runs)      # every run carries its page's bit width in kind >> 8
        bw = 0
        d_bytes = to_device(np.frombuffer(bytes(index_bytes) or b"\0", dtype=np.uint8), device)
        d_runs = _device_runs(runs, device)
        idx = alloc(dense_from_dict * 4, device)
        check(lib.arx_rle_decode_u32(d_bytes.data_ptr(), len(index_bytes), d_runs.data_ptr(), len(runs), bw,
                                     dense_from_dict, idx.data_ptr(), stream))
        d_dict = to_device(np.frombuffer(dict_bytes[: dict_count * width], dtype=np.uint8), device)
        dvals = Array(atype, dict_count, [None, d_dict], 0, 0)
        didx = Array(_lib_uint32(), dense_from_dict, [None, idx], 0, 0)
        from . import compute as cp

        part = cp.take(dvals, didx, boundscheck=True)               # a corrupt index fails like the reference's bounds check
        dense_buf[: dense_from_dict * width] = part.data[: dense_from_dict * width]
    if split_pages:
        # ByteStreamSplitDecoder: every page is `width` byte streams; one transposing launch per page
        d_split = to_device(np.frombuffer(b"".join(p[1] for p in split_pages), dtype=np.uint8), device)
        at = 0
        for start, data, count in split_pages:
            check(lib.arx_byte_stream_split_decode(d_split.data_ptr() + at, count, width, dense_buf.data_ptr() + start * width,
                                                   stream))
            at += len(data)
    if plain_pos:
        if delta_pages or split_pages:
            raise ArrowNotImplementedError("Parquet: PLAIN and DELTA_BINARY_PACKED / BYTE_STREAM_SPLIT pages in one column chunk")
        # only the HOST-decoded pages' bytes cross PCIe (zero placeholders for the device-decoded pages used to
        # ride along, the whole uncompressed size on top of the compressed blocks); adjacent segments go in one copy
        merged = []
        for pos_, data in plain_segments:
            if merged and merged[-1][0] + len(merged[-1][1]) == pos_:
                merged[-1][1] += data
            else:
                merged.append([pos_, bytearray(data)])
        for pos_, data in merged:
            host = torch.from_numpy(np.frombuffer(bytes(data), dtype=np.uint8).copy())
            at0 = dense_from_dict * width + pos_
            dense_buf[at0: at0 + len(data)] = host.to(device)
        if device_snappy_pages:
            # one launch for all the chunk's device-decoded pages: the compressed blocks cross PCIe, one wave per page
            # writes its values straight into the dense buffer (after the host-decoded pages' copy above)
            table = np.zeros(len(device_snappy_pages), SNAPPY_PAGE_DTYPE)
            at = 0
            for i, (blk, usize, pos_) in enumerate(device_snappy_pages):
                table[i] = (at, len(blk), usize, dense_from_dict * width + pos_)
                at += len(blk)
            d_src = to_device(np.frombuffer(b"".join(b for b, _, _ in device_snappy_pages) + b"\0", dtype=np.uint8), device)
            d_table = to_device(table.view(np.uint8), device)
            status = torch.zeros(len(device_snappy_pages), dtype=torch.int32, device=device)
            decompress = lib.arx_gzip_decompress_pages if codec == "GZIP" else lib.arx_snappy_decompress_pages
            check(decompress(d_src.data_ptr(), d_table.data_ptr(), len(device_snappy_pages),
                             dense_buf.data_ptr(), status.data_ptr(), stream))
            bad = status.cpu().numpy()
            if bad.any():
                name = "GZIP" if codec == "GZIP" else "Snappy"
                raise ArrowInvalid(f"Parquet: corrupt {name} page (device decoder status {int(bad[bad != 0][0])} on page "
                                   f"{int(np.nonzero(bad)[0][0])} of the chunk's device-decoded pages)")
            if stats is not None:
                key = "device_gzip_pages" if codec == "GZIP" else "device_snappy_pages"
                stats[key] = stats.get(key, 0) + len(device_snappy_pages)
    if delta_pages:
        # one byte buffer, one miniblock table and one page table for the chunk: ONE launch sequence (unpack + prefix
        # sum) for all its pages — every page restarts the recurrence at its own first value, so pages own whole tiles
        d_bytes = to_device(np.frombuffer(bytes(delta_bytes) + b"\0" * 16, dtype=np.uint8), device)
        table = np.concatenate([p[1] for p in delta_pages]) if any(len(p[1]) for p in delta_pages) else np.zeros(1, MINIBLOCK_DTYPE)
        d_table = to_device(table.view(np.uint8), device)
        pages = np.zeros(len(delta_pages), dtype=DELTA_PAGE_DTYPE)
        at = tiles = 0
        for j, (start, mbs, vpm, first, count) in enumerate(delta_pages):
            pages[j] = (start, at, max(vpm, 1), first, count, tiles)
            at += len(mbs)
            tiles += (count + 4095) // 4096
        d_pages = to_device(pages.view(np.uint8), device)
        ws = alloc(lib.arx_delta_decode_workspace_bytes(max(tiles, 1) * 4096), device)
        check(lib.arx_delta_decode_pages(d_bytes.data_ptr(), d_table.data_ptr(), d_pages.data_ptr(), len(delta_pages), tiles,
                                         width, ws.data_ptr(), ws.numel(), dense_buf.data_ptr(), stream))

    if max_def_level == 0 or dense == rows:                         # required column, or optional without a single null
        return Array(atype, rows, [None, dense_buf], 0, 0)
    # ---- validity bitmap from the definition levels, then spread the dense values over their slots
    validity, mask, ws = _validity_and_ws(lib, stream, device, level_bytes, level_runs, rows, dense, max_def_level)
    out = alloc(rows * width, device)
    check(lib.arx_expand_by_mask(dense_buf.data_ptr(), width, C.byref(mask), ws.data_ptr(), out.data_ptr(), stream))
    return Array(atype, rows, [validity, out], rows - dense, 0)


def _validity_and_ws(lib, stream, device, level_bytes, level_runs, rows, dense, max_def_level):
    """Definition levels -> (validity bitmap, mask span, count/scan workspace) for an optional column."""
    runs = np.concatenate(level_runs)
    d_lbytes = to_device(np.frombuffer(bytes(level_bytes) or b"\0", dtype=np.uint8), device)
    d_lruns = _device_runs(runs, device)
    validity = alloc(bitmap_nbytes(rows), device, zero=True)
    check(lib.arx_rle_decode_equals_bitmap(d_lbytes.data_ptr(), len(level_bytes), d_lruns.data_ptr(), len(runs),
                                           max(1, int(max_def_level).bit_length()), rows, max_def_level,
                                           validity.data_ptr(), stream))
    mask = _lib.ArxSpan(None, validity.data_ptr(), 0, rows, 0)
    ws = alloc(lib.arx_filter_workspace_bytes(rows) + 64, device)
    cnt = C.c_int64(0)
    check(lib.arx_filter_count(C.byref(mask), _lib.FILTER_DROP, ws.data_ptr(), ws.numel(), C.byref(cnt), stream))
    if cnt.value != dense:
        raise ArrowInvalid(f"Parquet: {cnt.value} non-null definition levels but {dense} values (corrupt page?)")
    return validity, mask, ws


def _finish_boolean_chunk(lib, stream, device, bool_bytes, bool_runs, level_bytes, level_runs, rows, dense,
                          max_def_level) -> Array:
    """BOOLEAN column: the pages' values (PLAIN bit-packing = a literal run, or RLE) are decoded by the
    hybrid decoder straight into a dense bitmap; an optional column then takes bit rank(r) for every valid
    slot r — the ranks are 0..dense-1 spread over the slots by the validity bitmap (arx_expand_by_mask),
    the gather is arx_take_bits."""
    from . import compute as cp
    from .array import bool_, uint32

    dense_bits = alloc(bitmap_nbytes(max(dense, 1)), device, zero=True)
    if dense:
        runs = np.concatenate(bool_runs)
        d_bytes = to_device(np.frombuffer(bytes(bool_bytes) or b"\0", dtype=np.uint8), device)
        d_runs = _device_runs(runs, device)
        check(lib.arx_rle_decode_equals_bitmap(d_bytes.data_ptr(), len(bool_bytes), d_runs.data_ptr(), len(runs), 1,
                                               dense, 1, dense_bits.data_ptr(), stream))
    if max_def_level == 0 or dense == rows:
        return Array(bool_, rows, [None, dense_bits], 0, 0)
    validity, mask, ws = _validity_and_ws(lib, stream, device, level_bytes, level_runs, rows, dense, max_def_level)
    ranks = torch.arange(max(dense, 1), dtype=torch.int32, device=device)
    full = alloc(rows * 4, device)
    check(lib.arx_expand_by_mask(ranks.data_ptr(), 4, C.byref(mask), ws.data_ptr(), full.data_ptr(), stream))
    didx = Array(uint32, rows, [validity, full], rows - dense, 0)
    return cp.take(Array(bool_, dense, [None, dense_bits], 0, 0), didx, boundscheck=False)


def _finish_binary_chunk(lib, stream, device, atype, dict_bytes, dict_count, index_bytes, index_runs, level_bytes,
                         level_runs, rows, dense, max_def_level, dense_from_dict, plain_pages) -> Array:
    """utf8 / binary column.  All values are gathered from ONE var-width "source" array by one take
    (arx_binary_take_*), which builds offsets, validity and bytes in HBM:
      * dictionary-encoded pages: source entries = the dictionary, indices = the decoded RLE indices;
      * PLAIN pages (also after a dictionary -> PLAIN fallback): the page bytes are appended to the
        source as alternating {4-byte length prefix, value} entries (offsets from one host walk,
        arx_plain_byte_array_offsets) and the indices are simply the odd entries;
      * DELTA_LENGTH_BYTE_ARRAY pages: the bytes are appended as they are, the entries' offsets are the running
        sum of the delta-decoded lengths (arx_delta_decode + arx_lengths_to_offsets_i32), indices consecutive.
    For an optional column the dense indices are first spread over their slots with the validity
    bitmap (which becomes the indices' validity)."""
    from . import compute as cp
    from .array import uint32

    offs, data = _parse_byte_array_dictionary(dict_bytes or b"", dict_count)
    # ---- the source array: [dictionary entries][entries of every non-dictionary page, in page order]
    # (DELTA_BYTE_ARRAY pages are expanded on the device: their entries — and bytes — come LAST in the source, whatever
    #  their place among the pages; the indices say which entry a slot reads)
    dba_pages = [(page, c) for page, c, kind in plain_pages if kind == "delta_byte_array"]
    dba_total = sum(c for _, c in dba_pages)
    total_entries = dict_count + sum(2 * c if kind == "plain" else c for _, c, kind in plain_pages)
    host_entries = total_entries - dba_total
    d_offs = alloc((total_entries + 1) * 4, device)
    d_offs[: (dict_count + 1) * 4] = to_device(offs.view(np.uint8), device)[: (dict_count + 1) * 4]
    data_parts, base, epos = [data], len(data), dict_count
    idx = alloc(max(dense, 1) * 4, device)
    ipos = dense_from_dict                                   # dense slot of the next non-dictionary value
    dba_epos = host_entries
    for page, count, kind in plain_pages:
        if kind == "delta_byte_array":
            if count:
                seq = torch.arange(dba_epos, dba_epos + count, dtype=torch.int32, device=device)
                idx[ipos * 4: (ipos + count) * 4] = seq.view(torch.uint8)
            dba_epos += count
            ipos += count
            continue
        if kind == "plain":
            # alternating {4-byte length prefix, value} entries; the values are the odd ones
            o = np.zeros(2 * count + 1, dtype=np.int32)
            buf = np.frombuffer(page, dtype=np.uint8)
            check(lib.arx_plain_byte_array_offsets(buf.ctypes.data if len(buf) else None, len(buf), count, base, o.ctypes.data))
            used = int(o[-1]) - base
            if count:
                d_offs[(epos + 1) * 4: (epos + 1 + 2 * count) * 4] = to_device(o[1:].view(np.uint8), device)[: 8 * count]
                seq = torch.arange(epos + 1, epos + 1 + 2 * count, 2, dtype=torch.int32, device=device)
                idx[ipos * 4: (ipos + count) * 4] = seq.view(torch.uint8)
            data_parts.append(page[:used])
            epos += 2 * count
        else:
            # DeltaLengthByteArrayDecoder (parquet/decoder.cc): DELTA_BINARY_PACKED lengths, then all the bytes; the
            # offsets are the running sum of the lengths, built on the device right where the source array wants them
            if count == 0:
                continue
            lengths = decode_delta_binary_packed(page, 4, device)
            if lengths.length != count:
                raise ArrowInvalid(f"Parquet: DELTA_LENGTH_BYTE_ARRAY page holds {lengths.length} lengths, its header says {count}")
            _, _, _, _, header_bytes = scan_delta_miniblocks(page)
            used = len(page) - header_bytes
            ws = alloc(lib.arx_delta_decode_workspace_bytes(count + 1), device)
            check(lib.arx_lengths_to_offsets_i32(lengths.data.data_ptr(), count, base, d_offs.data_ptr() + epos * 4,
                                                 ws.data_ptr(), ws.numel(), stream))
            seq = torch.arange(epos, epos + count, dtype=torch.int32, device=device)
            idx[ipos * 4: (ipos + count) * 4] = seq.view(torch.uint8)
            data_parts.append(page[header_bytes:])
            epos += count
        base += used
        ipos += count
    if base > 2**31 - 1:
        raise ArrowInvalid("Parquet: column chunk exceeds the int32 offset range")
    if any(kind == "delta_length" for _, _, kind in plain_pages):
        end = int(d_offs.view(torch.int32)[host_entries].item())
        if end != base:      # the lengths must add up to the bytes the pages carry
            raise ArrowInvalid(f"Parquet: DELTA_LENGTH_BYTE_ARRAY lengths sum to {end - len(data)}, the pages carry {base - len(data)} bytes")
    d_data = to_device(np.frombuffer(b"".join(data_parts) or b"\0", dtype=np.uint8), device)
    if dba_total:
        if host_entries == 0:
            d_offs.view(torch.int32)[0] = 0
        _, expanded, _ = expand_delta_byte_array(dba_pages, base, d_offs[host_entries * 4:], device)
        d_data = torch.cat([d_data[:base], expanded]) if expanded.numel() else d_data
    dvals = Array(atype, total_entries, [None, d_offs, d_data], 0, 0)
    if dense_from_dict:
        runs = np.concatenate(index_runs)
        d_bytes = to_device(np.frombuffer(bytes(index_bytes) or b"\0", dtype=np.uint8), device)
        d_runs = _device_runs(runs, device)
        check(lib.arx_rle_decode_u32(d_bytes.data_ptr(), len(index_bytes), d_runs.data_ptr(), len(runs), 0,
                                     dense_from_dict, idx.data_ptr(), stream))
    if max_def_level == 0 or dense == rows:
        didx = Array(uint32, rows, [None, idx], 0, 0)
        return cp.take(dvals, didx, boundscheck=True)
    validity, mask, ws = _validity_and_ws(lib, stream, device, level_bytes, level_runs, rows, dense, max_def_level)
    full = alloc(rows * 4, device)
    check(lib.arx_expand_by_mask(idx.data_ptr(), 4, C.byref(mask), ws.data_ptr(), full.data_ptr(), stream))
    didx = Array(uint32, rows, [validity, full], rows - dense, 0)
    return cp.take(dvals, didx, boundscheck=True)


def list_level_infos(field):
    """LevelInfo of every list level and of the leaf of a chain list<list<...<primitive>>> (the 3-level LIST encoding) —
    what the reference's schema walk computes (cpp/src/parquet/arrow/schema.cc ListToSchemaField, LevelInfo::Increment /
    IncrementRepeated, level_conversion.h:31-135).  Returns ([(def_level, rep_level, repeated_ancestor_def_level) per list
    level, outermost first], the leaf's triple, the leaf's pyarrow type)."""
    import pyarrow as pa

    d = r = anc = 0
    lists = []
    t, nullable = field.type, field.nullable
    while pa.types.is_list(t):
        if nullable:
            d += 1                      # the optional group that carries the LIST annotation
        d += 1                          # the repeated group: IncrementRepeated returns the ancestor level so far ...
        r += 1
        lists.append((d, r, anc))
        anc = d                         # ... and every descendant's repeated ancestor is this list
        t, nullable = t.value_field.type, t.value_field.nullable
    if pa.types.is_nested(t) or pa.types.is_large_list(t) or pa.types.is_fixed_size_list(t):
        raise ArrowNotImplementedError(f"Parquet: {field.type} columns are not on the gfx950 path (lists of primitives are)")
    if nullable:
        d += 1
    return lists, (d, r, anc), t


def assemble_list_column(leaf: Array, nested: dict, field, num_rows: int, device=None):
    """A list column from its leaf values and levels: DefRepLevelsToList per list level (outermost first; the entries of a
    level are the elements of the one above), the leaf's slots = the level slots whose def level reaches its repeated
    ancestor (a DROP filter of the values spread over all level slots).  The reference: parquet/arrow/reader.cc
    ListReader::BuildArray over level_conversion.cc:40-146."""
    from . import compute as cp
    from .array import ListArray, bool_

    device = torch.device(device) if device is not None else default_device()
    lib, stream = _lib.get_lib(), current_stream(device)
    lists, (leaf_def, _leaf_rep, leaf_anc), leaf_pa_type = list_level_infos(field)
    n = nested["levels"]
    try:        # logical types that share the physical layout (timestamp, date32, time32 / time64): labelled as the reference does
        from .array import is_temporal, type_from_name

        logical = type_from_name(str(leaf_pa_type))
        if is_temporal(logical) and getattr(leaf.type, "bit_width", 0) == logical.bit_width:
            leaf.type = logical
    except ArrowNotImplementedError:
        pass

    def decode(level_bytes, runs_list, bw):
        out = alloc(max(n, 1) * 4, device)
        if n == 0 or bw == 0:
            out.zero_()
            return out
        runs = np.concatenate(runs_list)
        d_bytes = to_device(np.frombuffer(level_bytes or b"\0", dtype=np.uint8), device)
        d_runs = _device_runs(runs, device)
        check(lib.arx_rle_decode_u32(d_bytes.data_ptr(), len(level_bytes), d_runs.data_ptr(), len(runs), bw, n,
                                     out.data_ptr(), stream))
        return out

    d_def = decode(nested["def_bytes"], nested["def_runs"], nested["def_bw"])
    d_rep = decode(nested["rep_bytes"], nested["rep_runs"], nested["rep_bw"])
    ws = alloc(lib.arx_levels_to_list_workspace_bytes(n), device)
    counts = torch.zeros(4, dtype=torch.int64, device=device)
    levels_out = []
    entries = num_rows
    for (dl, rl, anc) in lists:
        offsets = alloc((entries + 1) * 4, device)
        valid = alloc(bitmap_nbytes(max(entries, 1)), device, zero=True)
        check(lib.arx_def_rep_levels_to_list(d_def.data_ptr(), d_rep.data_ptr(), n, dl, rl, anc, entries, offsets.data_ptr(),
                                             valid.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        got, elems, nulls, over = (int(x) for x in counts.cpu().tolist())
        if over or got != entries:
            raise ArrowInvalid(f"Parquet: the levels hold {got} list entries where {entries} are expected (corrupt page?)")
        levels_out.append((entries, offsets, valid if nulls else None, nulls))
        entries = elems
    # the leaf: keep the level slots that exist below the innermost list
    keep = alloc(bitmap_nbytes(max(n, 1)), device, zero=True)
    check(lib.arx_levels_ge_bitmap(d_def.data_ptr(), n, leaf_anc, keep.data_ptr(), None, stream))
    child = cp.filter(leaf, Array(bool_, n, [None, keep], 0, 0))
    if child.length != entries:
        raise ArrowInvalid(f"Parquet: {child.length} leaf slots for {entries} list elements (corrupt page?)")
    ftype = field.type
    types = []
    while len(types) < len(lists):
        types.append(ftype)
        ftype = ftype.value_field.type
    for (length, offsets, valid, nulls), pa_type in zip(reversed(levels_out), reversed(types)):
        child = ListArray(pa_type, length, [valid, offsets], child, nulls)
    return child


def _read_struct_column(pf, md, raw, names, top: str, device, stats):
    """A struct column whose members are primitives: every member is a flat leaf with one more definition level when the
    struct is nullable (LevelInfo::Increment, level_conversion.h; no repetition), decoded with one slot per row — valid where
    def = its maximum level, which is what the reference's StructReader::BuildArray leaves in a member under a null struct —
    and the struct's own validity is DefLevelsToBitmap of ANY member's levels at the struct's level (def >= 1;
    parquet/arrow/reader.cc StructReader::GetDefLevels picks the first child with levels).  One StructArray per row group."""
    import pyarrow as pa

    from .array import StructArray, binary, is_temporal, type_from_name, utf8

    field = pf.schema_arrow.field(top)
    members = [i for i, nm in enumerate(names) if nm.split(".")[0] == top]
    if not pa.types.is_struct(field.type) or len(members) != field.type.num_fields or \
            any(pa.types.is_nested(field.type.field(j).type) for j in range(field.type.num_fields)):
        raise ArrowNotImplementedError(f"Parquet: column {top} of type {field.type} is not on the gfx950 path "
                                       "(structs of primitives and lists of primitives are)")
    device = torch.device(device) if device is not None else default_device()
    lib, stream = _lib.get_lib(), current_stream(device)
    struct_def = 1 if field.nullable else 0
    chunks = []
    for rg in range(md.num_row_groups):
        rows = md.row_group(rg).num_rows
        children, validity, nulls = [], None, 0
        for j, ci in enumerate(members):
            col = md.schema.column(ci)
            mfield = field.type.field(j)
            want_def = struct_def + (1 if mfield.nullable else 0)
            if col.max_definition_level != want_def or col.max_repetition_level:
                raise ArrowNotImplementedError(f"Parquet: level layout of column {names[ci]}")
            binary_type = None
            if col.physical_type == "BYTE_ARRAY":
                binary_type = utf8 if str(col.logical_type).upper().startswith("STRING") else binary
            nested = {"max_rep": 0}
            child = read_column_chunk(raw, md.row_group(rg).column(ci), want_def, device, stats, binary_type, nested)
            if child.length != rows:
                raise ArrowInvalid(f"Parquet: {child.length} levels in column {names[ci]} for {rows} rows (corrupt page?)")
            if col.physical_type in ("INT32", "INT64"):
                try:
                    logical = type_from_name(str(mfield.type))
                except ArrowNotImplementedError:
                    logical = None
                if logical is not None and is_temporal(logical) and child.type.bit_width == logical.bit_width:
                    child.type = logical
            children.append(child)
            if struct_def and validity is None and want_def > 0 and nested["levels"] == rows:
                # the struct's validity from this member's levels: def >= the struct's level
                runs = np.concatenate(nested["def_runs"]) if nested["def_runs"] else np.zeros(0, RUN_DTYPE)
                d_def = alloc(max(rows, 1) * 4, device)
                if rows and len(runs):
                    d_bytes = to_device(np.frombuffer(nested["def_bytes"] or b"\0", dtype=np.uint8), device)
                    d_runs = _device_runs(runs, device)
                    check(lib.arx_rle_decode_u32(d_bytes.data_ptr(), len(nested["def_bytes"]), d_runs.data_ptr(), len(runs),
                                                 nested["def_bw"], rows, d_def.data_ptr(), stream))
                else:
                    d_def.zero_()
                validity = alloc(bitmap_nbytes(max(rows, 1)), device, zero=True)
                ones = torch.zeros(1, dtype=torch.int64, device=device)
                check(lib.arx_levels_ge_bitmap(d_def.data_ptr(), rows, struct_def, validity.data_ptr(), ones.data_ptr(), stream))
                nulls = rows - int(ones.item())
                if nulls == 0:
                    validity = None
        chunks.append(StructArray(field.type, rows, [validity], children, nulls))
    return chunks


def _lib_uint32():
    from .array import uint32

    return uint32


def read_table(path: str, columns=None, device=None, stats: dict | None = None) -> dict:
    """{column name: [device array per row group]}: flat columns by their name, list columns by the path of their leaf,
    struct columns (of primitives) by the name of their top-level field."""
    import pyarrow.parquet as pq

    pf = pq.ParquetFile(path)
    md = pf.metadata
    with open(path, "rb") as f:
        raw = f.read()
    names = [md.schema.column(i).path for i in range(md.num_columns)]
    wanted = names if columns is None else list(columns)
    out = {}
    for name in wanted:
        if name not in names:            # a nested column by the name of its top-level field: (the first of) its leaves
            under = [nm for nm in names if nm.split(".")[0] == name]
            if not under:
                raise ArrowInvalid(f"Parquet: no column {name!r} in {path}")
            name = under[0]
        ci = names.index(name)
        max_def = md.schema.column(ci).max_definition_level
        max_rep = md.schema.column(ci).max_repetition_level
        binary_type = None
        if md.schema.column(ci).physical_type == "BYTE_ARRAY":
            from .array import binary, utf8

            binary_type = utf8 if str(md.schema.column(ci).logical_type).upper().startswith("STRING") else binary
        if not max_rep and "." in name:
            # a member of a struct column: the whole struct is assembled once, under the name of its top-level field
            top = name.split(".")[0]
            if top not in out:
                out[top] = _read_struct_column(pf, md, raw, names, top, device, stats)
            continue
        if max_rep:
            # a repeated column: lists (of lists ...) of a primitive, addressed by the name of its top-level field
            top = name.split(".")[0]
            field = pf.schema_arrow.field(top)
            lists, leaf_info, _ = list_level_infos(field)
            if not lists or leaf_info[0] != max_def or leaf_info[1] != max_rep or \
                    sum(1 for nm in names if nm.split(".")[0] == top) != 1:
                raise ArrowNotImplementedError(f"Parquet: column {name} is not a 3-level LIST chain over one primitive")
            chunks = []
            for rg in range(md.num_row_groups):
                nested = {"max_rep": max_rep}
                leaf = read_column_chunk(raw, md.row_group(rg).column(ci), max_def, device, stats, binary_type, nested)
                chunks.append(assemble_list_column(leaf, nested, field, md.row_group(rg).num_rows, device))
            out[name] = chunks
            continue
        chunks = [read_column_chunk(raw, md.row_group(rg).column(ci), max_def, device, stats, binary_type)
                  for rg in range(md.num_row_groups)]
        # logical types whose Arrow layout is the physical layout (timestamp, date32, time32 / time64): same bytes,
        # labelled the way the reference's reader labels them (parquet/arrow/schema.cc)
        if md.schema.column(ci).physical_type in ("INT32", "INT64") and "." not in name:
            from .array import is_temporal, type_from_name

            try:
                logical = type_from_name(str(pf.schema_arrow.field(name).type))
            except ArrowNotImplementedError:
                logical = None
            if logical is not None and is_temporal(logical):
                for a in chunks:
                    if a.type.bit_width == logical.bit_width:
                        a.type = logical
        out[name] = chunks
    return out
