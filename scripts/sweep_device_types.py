"""Which (function, value type) pairs does a device-resident array survive?

Every call runs in a forked child (a CPU kernel handed HBM pointers dies with SIGSEGV inside libarrow), so one run prints
the whole map: `ok`, `refused:<error class>` or `CRASH(sig N)` per pair.  Round 3 used it to find the pairs that are now
served on the device (divide on every numeric type, 8- / 16-bit sort keys, uint64 extrema, mean, is_valid / is_null) or
refused by the guards of plugin/device_guard.inc; round 4 closed the rest of its list (fill_null, the scalar aggregates of
float / boolean / temporal columns: served; decimals: refused by name) — the map printed at the end of round 4 has no CRASH cell.

    ARROW_AMD_PLUGIN_EMULATED=1 python scripts/sweep_device_types.py [function ...]     # GPU-less: the emulated shim
    python scripts/sweep_device_types.py                                               # on an MI355X: the real one
"""
import ctypes, os, sys, signal, decimal
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, pyarrow as pa, pyarrow.compute as pc
if os.environ.get("ARROW_AMD_PLUGIN_EMULATED") == "1":
    from tests.emu.build_plugin_emu import build_plugin
else:
    from arrow_amd.plugin_build import build_plugin
path = build_plugin()
lib = ctypes.CDLL(path)
lib.arrow_amd_plugin_last_error.restype = ctypes.c_char_p
assert lib.arrow_amd_register() == 0
lib.arrow_amd_plugin_set_min_rows(ctypes.c_int64(0))
def to_device(arr):
    c_arr, c_schema, c_dev = (ctypes.create_string_buffer(m) for m in (80, 72, 128))
    arr._export_to_c(ctypes.addressof(c_arr), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_device(c_arr, c_schema, c_dev) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c_device(ctypes.addressof(c_dev), arr.type)
def to_host(x):
    if isinstance(x, pa.Scalar) or isinstance(x, pa.ChunkedArray): return x
    if isinstance(x, pa.StructArray):
        return x
    if all(b is None or b.is_cpu for b in x.buffers()): return x
    c_dev, c_schema, c_arr = ctypes.create_string_buffer(128), ctypes.create_string_buffer(72), ctypes.create_string_buffer(80)
    x._export_to_c_device(ctypes.addressof(c_dev), ctypes.addressof(c_schema))
    assert lib.arrow_amd_copy_to_host(c_dev, c_schema, c_arr, None) == 0, lib.arrow_amd_plugin_last_error()
    return pa.Array._import_from_c(ctypes.addressof(c_arr), x.type)
base=[3,1,None,2,3,0,None,1]
types={"bool":pa.array([True,False,None,True,True,False,None,False]),
 **{n:pa.array(base,getattr(pa,n)()) for n in ("int8","uint8","int16","uint16","int32","uint32","int64","uint64","float32","float64")},
 "date32":pa.array(base,pa.int32()).cast(pa.date32()),"date64":pa.array(base,pa.int64()).cast(pa.date64()),
 "timestamp[us]":pa.array(base,pa.int64()).cast(pa.timestamp("us")),"duration[s]":pa.array(base,pa.int64()).cast(pa.duration("s")),
 "time32[ms]":pa.array(base,pa.int32()).cast(pa.time32("ms")),"time64[ns]":pa.array(base,pa.int64()).cast(pa.time64("ns")),
 "string":pa.array(["c","a",None,"b","c","",None,"a"]),"binary":pa.array([b"c",b"a",None,b"b",b"c",b"",None,b"a"]),
 "large_string":pa.array(["c","a",None,"b","c","",None,"a"],pa.large_string()),
 "decimal128":pa.array([decimal.Decimal(x) if x is not None else None for x in base],pa.decimal128(10,2)),
 "fixed_size_binary":pa.array([b"cc",b"aa",None,b"bb",b"cc",b"zz",None,b"aa"],pa.binary(2)),
 "float16":pa.array(np.array([3,1,0,2,3,0,0,1],dtype=np.float16),mask=np.array([0,0,1,0,0,0,1,0],bool))}
mask=pa.array([True,False,True,None,True,True,False,True]); idx=pa.array([7,0,None,3,3],pa.int32())
fns={"filter":lambda a,d:pc.filter(d,to_device(mask)),"filter_emit":lambda a,d:pc.filter(d,to_device(mask),null_selection_behavior="emit_null"),
 "take":lambda a,d:pc.take(d,to_device(idx)),"drop_null":lambda a,d:pc.drop_null(d),"unique":lambda a,d:pc.unique(d),
 "value_counts":lambda a,d:pc.value_counts(d),"dictionary_encode":lambda a,d:pc.dictionary_encode(d),
 "array_sort_indices":lambda a,d:pc.array_sort_indices(d),"sort_indices":lambda a,d:pc.sort_indices(d),
 "count":lambda a,d:pc.count(d),"sum":lambda a,d:pc.sum(d),"mean":lambda a,d:pc.mean(d),"min_max":lambda a,d:pc.min_max(d),
 "equal":lambda a,d:pc.equal(d,d),"greater":lambda a,d:pc.greater(d,d),"add":lambda a,d:pc.add(d,d),"multiply":lambda a,d:pc.multiply(d,d),"divide":lambda a,d:pc.divide(d,d),
 "is_null":lambda a,d:pc.is_null(d),"is_valid":lambda a,d:pc.is_valid(d),"fill_null":lambda a,d:pc.fill_null(d,a[0]),"cast_self":lambda a,d:pc.cast(d,a.type),
 "indices_nonzero":lambda a,d:pc.indices_nonzero(d)}
only=sys.argv[1:] 
res={}
for fn,f in fns.items():
    if only and fn not in only: continue
    for tn,arr in types.items():
        r,w=os.pipe(); pid=os.fork()
        if pid==0:
            os.close(r); out="?"
            try:
                got=f(arr,to_device(arr)); out="ok"
            except (pa.ArrowNotImplementedError,pa.ArrowInvalid,pa.ArrowTypeError,TypeError) as e:
                out="refused:"+type(e).__name__
            except Exception as e:
                out="exc:"+repr(e)[:80]
            os.write(w,out.encode()); os._exit(0)
        os.close(w); data=os.read(r,200).decode(); os.close(r); _,st=os.waitpid(pid,0)
        res[(fn,tn)]="CRASH(sig%d)"%(st&0x7f) if (st&0x7f) else data
for fn in fns:
    if only and fn not in only: continue
    row=[f"{tn}={res[(fn,tn)]}" for tn in types if res[(fn,tn)]!="ok"]
    print(fn, "ALL OK" if not row else "; ".join(row))
