// arrow_plugin.cc — the reference-side host shim: registers MI355X kernels on Arrow's OWN live
// FunctionRegistry so that CallFunction("filter" | "take" | "greater" | "array_sort_indices", ...)
// — and therefore pyarrow.compute, Acero, and every other caller — dispatches to the HIP kernels
// of libarrow_amd.so through the C ABI of include/arrow_amd.h.  Compiled with g++ against the
// installed Arrow (headers + libarrow.so.2500 of the pyarrow wheel); no Arrow source is patched.
//
// Mechanism (verified in SURVEY.md Appendix C): Function::AddKernel on the live function; the
// LAST matching kernel wins in DispatchExactImpl (cpp/src/arrow/compute/function.cc:122-157).
// Each added kernel is a COPY of the stock kernel (same NullHandling / MemAllocation /
// chunking flags, cpp/src/arrow/compute/kernel.h:561-660) with `signature`, `init` and `exec`
// replaced.  `init` chains to the stock init so the stock state object exists; `exec` hands
// shapes this shim does not cover (run-end-encoded filters, boolean values, scalars, tiny
// inputs, ...) to the stock exec with that state, exactly as the reference would have run them.
//
// Two kinds of input reach these kernels.  HOST arrays (ArraySpan::buffers[i].data, cpp/src/arrow/array/data.h:525-553)
// are staged through HBM over PCIe where that pays (filter, take, sort; thresholds in plugin/common.inc) and handed
// to the stock kernel otherwise.  DEVICE-RESIDENT arrays — buffers on the kROCM arrow::Device / MemoryManager of
// plugin/device.inc (interfaces: cpp/src/arrow/device.h:43-280), entering and leaving through the C Device Data
// interface — are computed in place and their outputs stay in HBM; that is the mode the
// numbers in DESIGN.md are quoted for and the one whole Acero plans run in (rows f2-f4: exec-node factories,
// sibling kernels, Parquet decode).
#include <arrow/acero/exec_plan.h>
#include <arrow/acero/options.h>
#include <arrow/api.h>
#include <arrow/c/abi.h>
#include <arrow/c/bridge.h>
#include <arrow/c/dlpack_abi.h>
#include <arrow/compute/api.h>
#include <arrow/compute/initialize.h>
#include <arrow/compute/kernel.h>
#include <arrow/compute/registry.h>
#include <arrow/device.h>
#include <arrow/util/bit_util.h>
#include <arrow/util/bitmap_builders.h>
#include <arrow/util/bitmap_ops.h>
#include <arrow/acero/util.h>
#include <arrow/acero/query_context.h>

#include <hip/hip_runtime_api.h>
#include <parquet/column_page.h>
#include <parquet/column_reader.h>
#include <parquet/file_reader.h>
#include <parquet/metadata.h>
#include <parquet/schema.h>
#include <parquet/arrow/schema.h>
#include <arrow/util/ubsan.h>
#include <arrow/util/compression.h>
#include <arrow/io/file.h>
#include <arrow/io/memory.h>
#include <arrow/io/interfaces.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <dlfcn.h>

#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/arrow_amd.h"

namespace cp = arrow::compute;
using arrow::ArrayData;
using arrow::ArraySpan;
using arrow::Buffer;
using arrow::Status;
using arrow::Type;

namespace {
#include "plugin/common.inc"
#include "plugin/device.inc"
#include "plugin/selection.inc"
#include "plugin/selection_nested.inc"
#include "plugin/selection_meta.inc"
#include "plugin/scalar.inc"
#include "plugin/sort.inc"
#include "plugin/cast.inc"
#include "plugin/hash_aggregate.inc"
#include "plugin/hash_aggregate_more.inc"
#include "plugin/hash_aggregate_bool.inc"
#include "plugin/vector_hash.inc"
#include "plugin/scalar_aggregate.inc"
#include "plugin/coalesce.inc"
#include "plugin/acero_node.inc"
#include "plugin/acero_node_general.inc"
#include "plugin/sharded.inc"
#include "plugin/sharded_sort.inc"
#include "plugin/order_by_node.inc"
#include "plugin/rank.inc"
#include "plugin/acero_source.inc"
#include "plugin/acero_coalesce.inc"
#include "plugin/acero_override.inc"
#include "plugin/parquet.inc"
#include "plugin/parquet_nested.inc"
#include "plugin/device_guard.inc"
#include "plugin/validity.inc"
#include "plugin/registration.inc"

}  // namespace

extern "C" {

// Registers the MI355X kernels on Arrow's global registry (idempotent).  0 on success.
int arrow_amd_register(void) {
  std::call_once(g_once, [] { g_register_status = RegisterAll(); });
  if (!g_register_status.ok()) {
    t_error = g_register_status.ToString();
    return -1;
  }
  return 0;
}
const char* arrow_amd_plugin_last_error(void) { return t_error.c_str(); }
int64_t arrow_amd_plugin_gpu_calls(void) { return g_gpu_calls.load(); }
int64_t arrow_amd_plugin_stock_calls(void) { return g_stock_calls.load(); }
// Calls of `function` ("array_filter", "array_take", "greater", "array_sort_indices", "cast",
// "hash_sum", "add", "boolean" = and_kleene / or_kleene / invert, "compare" = equal / not_equal /
// greater_equal / less / less_equal) that ran on the GPU (gpu != 0) or were handed to the stock CPU kernel; -1 = unknown name.
int64_t arrow_amd_plugin_calls(const char* function, int gpu) {
  for (int i = 0; i < kNumFn; ++i) {
    if (std::strcmp(function, kFnNames[i]) == 0) return (gpu ? g_fn_gpu[i] : g_fn_stock[i]).load();
  }
  return -1;
}
// Host array (C Data interface, consumed) -> the same array with its buffers in HBM, exported
// through the C Device Data interface (device_type = ARROW_DEVICE_ROCM).  Fixed-width, boolean
// and binary / utf8 arrays, and nested arrays of those (run_end_encoded, struct).  0 on success.
int arrow_amd_copy_to_device(struct ArrowArray* in, struct ArrowSchema* schema, struct ArrowDeviceArray* out) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto host, arrow::ImportArray(in, schema));
    ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
    // buffers of the array and of its children (run_end_encoded: run ends + run values; struct); no dictionaries
    std::function<arrow::Result<std::shared_ptr<ArrayData>>(const ArrayData&)> upload =
        [&](const ArrayData& h) -> arrow::Result<std::shared_ptr<ArrayData>> {
      if (h.dictionary != nullptr || h.buffers.size() > 3) {
        return Status::NotImplemented("arrow_amd_copy_to_device: ", h.type->ToString());
      }
      std::vector<std::shared_ptr<Buffer>> bufs(h.buffers.size());
      for (size_t i = 0; i < h.buffers.size(); ++i) {
        if (h.buffers[i] != nullptr) {
          ARROW_ASSIGN_OR_RAISE(bufs[i], arrow::MemoryManager::CopyBuffer(h.buffers[i], mm));
        }
      }
      // null_count must be exact: nothing may popcount a device bitmap on the CPU later
      auto data = ArrayData::Make(h.type, h.length, std::move(bufs), h.GetNullCount(), h.offset);
      for (const auto& child : h.child_data) {
        ARROW_ASSIGN_OR_RAISE(auto c, upload(*child));
        data->child_data.push_back(std::move(c));
      }
      return data;
    };
    ARROW_ASSIGN_OR_RAISE(auto data, upload(*host->data()));
    return arrow::ExportDeviceArray(*arrow::MakeArray(data), nullptr, out);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// Device memory the CALLER owns (a torch tensor, a hipMalloc'd block) as an Arrow array, without a copy: a fixed-width or
// boolean array of `schema`'s type (consumed) over `data` (+ `validity`, NULL = no nulls) in HBM, exported through the C
// Device Data interface.  The buffers are borrowed: the memory must outlive every array made from the export (results
// of kernels are new buffers).  null_count < 0: counted here (arx_bitmap_popcount), because nothing may count a device
// bitmap on the CPU later.  0 on success.
int arrow_amd_wrap_device_memory(struct ArrowSchema* schema, int64_t length, int64_t null_count, const void* validity,
                                 const void* data, struct ArrowDeviceArray* out) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto type, arrow::ImportType(schema));
    const int width = type->id() == Type::BOOL ? 0 : FixedByteWidth(*type);
    if (type->id() != Type::BOOL && width == 0) return Status::NotImplemented("arrow_amd_wrap_device_memory: ", type->ToString());
    if (length < 0 || (length > 0 && data == nullptr)) return Status::Invalid("arrow_amd_wrap_device_memory: bad length / data");
    ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
    const int64_t data_bytes = width == 0 ? arrow::bit_util::BytesForBits(length) : length * width;
    auto values = std::make_shared<Buffer>(static_cast<const uint8_t*>(data), data_bytes, mm);
    std::shared_ptr<Buffer> bitmap;
    if (validity != nullptr) {
      bitmap = std::make_shared<Buffer>(static_cast<const uint8_t*>(validity), arrow::bit_util::BytesForBits(length), mm);
      if (null_count < 0) {
        hipStream_t st;
        ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
        ARROW_ASSIGN_OR_RAISE(null_count, DeviceNullCount(*bitmap, length, st));
      }
      if (null_count == 0) bitmap = nullptr;
    } else {
      null_count = 0;
    }
    auto arr = arrow::MakeArray(ArrayData::Make(std::move(type), length, {std::move(bitmap), std::move(values)}, null_count));
    return arrow::ExportDeviceArray(*arr, nullptr, out);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// Device array (C Device Data interface, consumed) -> host array (C Data interface).
int arrow_amd_copy_to_host(struct ArrowDeviceArray* in, struct ArrowSchema* schema, struct ArrowArray* out,
                           struct ArrowSchema* out_schema) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto dev, arrow::ImportDeviceArray(in, schema));
    // buffers, children (struct: value_counts) and dictionary (dictionary_encode) alike
    std::function<arrow::Result<std::shared_ptr<ArrayData>>(const ArrayData&)> to_host =
        [&](const ArrayData& d) -> arrow::Result<std::shared_ptr<ArrayData>> {
      std::vector<std::shared_ptr<Buffer>> bufs(d.buffers.size());
      for (size_t i = 0; i < bufs.size(); ++i) {
        if (d.buffers[i] != nullptr) {
          ARROW_ASSIGN_OR_RAISE(bufs[i], arrow::MemoryManager::CopyBuffer(d.buffers[i], arrow::default_cpu_memory_manager()));
        }
      }
      auto host = ArrayData::Make(d.type, d.length, std::move(bufs), d.null_count.load(), d.offset);
      for (const auto& child : d.child_data) {
        ARROW_ASSIGN_OR_RAISE(auto hc, to_host(*child));
        host->child_data.push_back(std::move(hc));
      }
      if (d.dictionary != nullptr) {
        ARROW_ASSIGN_OR_RAISE(host->dictionary, to_host(*d.dictionary));
      }
      return host;
    };
    ARROW_ASSIGN_OR_RAISE(auto data, to_host(*dev->data()));
    return arrow::ExportArray(*arrow::MakeArray(data), out, out_schema);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// ---- DLPack (cpp/src/arrow/c/dlpack.h:45-74 ExportArray / ExportDevice, which stop at CPU memory): a device-resident
// primitive array without nulls as a DLManagedTensor on {kDLROCM, device id} — zero-copy into PyTorch-ROCm
// (torch.from_dlpack of a "dltensor" capsule around *out).  The tensor keeps the array's buffers alive; its deleter
// drops them.  Same refusals as the reference: nulls, non-numeric types.
int arrow_amd_export_dlpack(struct ArrowDeviceArray* in, struct ArrowSchema* schema, void** out_managed_tensor) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto arr, arrow::ImportDeviceArray(in, schema));
    const ArrayData& d = *arr->data();
    if (!DataOnRocm(d)) return Status::Invalid("arrow_amd_export_dlpack: not a device-resident array");
    if (d.buffers[0] != nullptr && d.null_count.load() != 0) {
      return Status::TypeError("Can only use DLPack on arrays with no nulls.");   // (c/dlpack.cc's message)
    }
    DLDataType dt{};
    dt.lanes = 1;
    switch (d.type->id()) {
      case Type::INT8: case Type::INT16: case Type::INT32: case Type::INT64: dt.code = kDLInt; break;
      case Type::UINT8: case Type::UINT16: case Type::UINT32: case Type::UINT64: dt.code = kDLUInt; break;
      case Type::HALF_FLOAT: case Type::FLOAT: case Type::DOUBLE: dt.code = kDLFloat; break;
      default: return Status::TypeError("DataType is not compatible with DLPack spec: ", d.type->ToString());
    }
    const int width = FixedByteWidth(*d.type);
    dt.bits = static_cast<uint8_t>(8 * width);
    struct Ctx {
      std::shared_ptr<arrow::Array> array;
      int64_t shape;
      DLManagedTensor tensor;
    };
    auto* ctx = new Ctx{arr, d.length, {}};
    DLTensor& t = ctx->tensor.dl_tensor;
    // the slice offset goes into the pointer and byte_offset stays 0, as the reference's exporter does
    // (c/dlpack.cc:108-116) — torch.from_dlpack refuses a non-zero byte_offset
    t.data = reinterpret_cast<void*>(d.buffers[1]->address() + static_cast<uint64_t>(d.offset) * width);
    t.byte_offset = 0;
    t.device = DLDevice{kDLROCM, static_cast<int32_t>(d.buffers[1]->device()->device_id())};
    t.ndim = 1;
    t.dtype = dt;
    t.shape = &ctx->shape;
    t.strides = nullptr;
    ctx->tensor.manager_ctx = ctx;
    ctx->tensor.deleter = [](DLManagedTensor* self) { delete static_cast<Ctx*>(self->manager_ctx); };
    *out_managed_tensor = &ctx->tensor;
    return Status::OK();
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// ---- f1 self-checks reachable from the tests (the classes themselves are only visible through Arrow's interfaces):
// a stream + an event of the kROCM device: record the event behind a device-to-device copy on the stream, hand the
// copy out with the event attached (ArrowDeviceArray.sync_event), i.e. what a producer that does NOT synchronise does.
int arrow_amd_copy_on_stream_with_event(struct ArrowDeviceArray* in, struct ArrowSchema* schema, struct ArrowDeviceArray* out,
                                        struct ArrowSchema* out_schema) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto arr, arrow::ImportDeviceArray(in, schema));
    const ArrayData& d = *arr->data();
    if (!DataOnRocm(d) || d.buffers.size() != 2) return Status::Invalid("a device-resident primitive array, please");
    ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
    ARROW_ASSIGN_OR_RAISE(auto stream, mm->device()->MakeStream());
    ARROW_ASSIGN_OR_RAISE(auto event, mm->MakeDeviceSyncEvent());
    hipStream_t st = *static_cast<const hipStream_t*>(stream->get_raw());
    std::vector<std::shared_ptr<Buffer>> bufs(2);
    for (int i = 0; i < 2; ++i) {
      if (d.buffers[i] == nullptr) continue;
      ARROW_ASSIGN_OR_RAISE(auto fresh, AllocDevice(d.buffers[i]->size()));
      ARROW_RETURN_NOT_OK(FromArx(arx_buffer_copy(reinterpret_cast<const void*>(d.buffers[i]->address()),
                                                  reinterpret_cast<void*>(fresh->mutable_address()), d.buffers[i]->size(), st)));
      bufs[i] = std::move(fresh);
    }
    ARROW_RETURN_NOT_OK(event->Record(*stream));
    auto copy = ArrayData::Make(d.type, d.length, std::move(bufs), d.null_count.load(), d.offset);
    ARROW_RETURN_NOT_OK(arrow::ExportType(*d.type, out_schema));
    ARROW_RETURN_NOT_OK(arrow::ExportDeviceArray(*arrow::MakeArray(copy), event, out));
    // the stream may go: the event was recorded, whoever imports the array waits on it
    return stream->Synchronize();
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// stream waits the shims issued for imported buffers' sync events (DeviceSpan)
int64_t arrow_amd_plugin_sync_event_waits(void) { return g_sync_event_waits.load(); }
// MemoryManager::GetBufferWriter / GetBufferReader round trip: host bytes -> a fresh device buffer -> host bytes
int arrow_amd_device_buffer_round_trip(const void* src, int64_t nbytes, void* dst) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
    ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> buf, mm->AllocateBuffer(nbytes));
    ARROW_ASSIGN_OR_RAISE(auto writer, mm->GetBufferWriter(buf));
    const int64_t half = nbytes / 2;
    ARROW_RETURN_NOT_OK(writer->Write(src, half));
    ARROW_RETURN_NOT_OK(writer->Write(static_cast<const uint8_t*>(src) + half, nbytes - half));
    ARROW_ASSIGN_OR_RAISE(const int64_t told, writer->Tell());
    if (told != nbytes) return Status::Invalid("writer position ", told, " after ", nbytes, " bytes");
    if (writer->Write(src, 1).ok()) return Status::Invalid("a write past the end of the buffer succeeded");
    ARROW_RETURN_NOT_OK(writer->Close());
    ARROW_ASSIGN_OR_RAISE(auto reader, mm->GetBufferReader(buf));
    ARROW_ASSIGN_OR_RAISE(const int64_t size, reader->GetSize());
    if (size != nbytes) return Status::Invalid("reader size ", size);
    ARROW_ASSIGN_OR_RAISE(const int64_t got, reader->ReadAt(half, nbytes, static_cast<uint8_t*>(dst) + half));   // clamped
    if (got != nbytes - half) return Status::Invalid("ReadAt returned ", got);
    ARROW_ASSIGN_OR_RAISE(auto head, reader->Read(half));
    if (head->size() != half) return Status::Invalid("Read returned ", head->size());
    std::memcpy(dst, head->data(), static_cast<size_t>(half));
    return reader->Close();
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// ---- the sharded hash_sum group-by over RCCL (plugin/sharded.inc): one process per GPU.
// Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others by whatever channel the application has.
int arrow_amd_sharded_unique_id(void* out_128_bytes) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(const RcclApi* api, Rccl());
    RCCL_RETURN_NOT_OK(api, api->GetUniqueId(static_cast<NcclId*>(out_128_bytes)));
    return Status::OK();
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
int arrow_amd_sharded_comm_create(const void* id_128_bytes, int world, int rank, void** out_comm) {
  auto run = [&]() -> Status {
    if (world < 1 || world > 1024 || rank < 0 || rank >= world) return Status::Invalid("bad world / rank");
    ARROW_ASSIGN_OR_RAISE(const RcclApi* api, Rccl());
    NcclId id;
    std::memcpy(&id, id_128_bytes, sizeof(id));
    auto c = std::make_unique<ShardedComm>();
    c->world = world;
    c->rank = rank;
    RCCL_RETURN_NOT_OK(api, api->CommInitRank(&c->comm, world, id, rank));
    *out_comm = c.release();
    return Status::OK();
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
void arrow_amd_sharded_comm_destroy(void* comm) {
  auto* c = static_cast<ShardedComm*>(comm);
  if (c == nullptr) return;
  auto api = Rccl();
  if (api.ok() && c->comm != nullptr) (void)(*api)->CommDestroy(c->comm);
  delete c;
}
// This rank's row shard (device arrays, C Device Data interface, consumed) -> this rank's groups.  exchange: 0 = the
// partial aggregates (default), 1 = the rows.  stage_ms: NULL, or 5 doubles (consume / export / exchange / merge /
// finalize; rows: partition_rows / exchange / consume / finalize / 0) — then the stream is synchronised per stage.
int arrow_amd_sharded_group_by_sum(void* comm, struct ArrowDeviceArray* keys, struct ArrowSchema* keys_schema,
                                   struct ArrowDeviceArray* values, struct ArrowSchema* values_schema, int skip_nulls,
                                   uint32_t min_count, int exchange, struct ArrowDeviceArray* out_keys,
                                   struct ArrowSchema* out_keys_schema, struct ArrowDeviceArray* out_sums,
                                   struct ArrowSchema* out_sums_schema, double* stage_ms) {
  auto run = [&]() -> Status {
    if (comm == nullptr) return Status::Invalid("arrow_amd_sharded_group_by_sum: no communicator");
    ARROW_ASSIGN_OR_RAISE(auto k, arrow::ImportDeviceArray(keys, keys_schema));
    ARROW_ASSIGN_OR_RAISE(auto v, arrow::ImportDeviceArray(values, values_schema));
    if (stage_ms != nullptr) std::fill(stage_ms, stage_ms + 5, 0.0);
    std::shared_ptr<ArrayData> ok, os;
    ARROW_RETURN_NOT_OK(ShardedGroupBySum(*static_cast<ShardedComm*>(comm), *k->data(), *v->data(), skip_nulls != 0, min_count,
                                          exchange, &ok, &os, stage_ms));
    ARROW_RETURN_NOT_OK(arrow::ExportType(*ok->type, out_keys_schema));
    ARROW_RETURN_NOT_OK(arrow::ExportDeviceArray(*arrow::MakeArray(ok), nullptr, out_keys));
    ARROW_RETURN_NOT_OK(arrow::ExportType(*os->type, out_sums_schema));
    return arrow::ExportDeviceArray(*arrow::MakeArray(os), nullptr, out_sums);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// The caching device allocator: bytes parked on free lists, allocations served from them / by hipMalloc.

// This rank's slice of array_sort_indices over the row shards of all ranks (uint64 / int64 keys): out_indices = global
// row numbers (uint64, device-resident), *out_start = where the slice begins in the global result; stage_ms (4 doubles or
// NULL): histogram / partition / exchange / local sort.
int arrow_amd_sharded_sort_indices(void* comm, struct ArrowDeviceArray* values, struct ArrowSchema* values_schema, int descending,
                                   int nulls_first, int splitter_bits, struct ArrowDeviceArray* out_indices,
                                   struct ArrowSchema* out_schema, int64_t* out_start, double* stage_ms) {
  auto run = [&]() -> Status {
    if (comm == nullptr || out_start == nullptr) return Status::Invalid("arrow_amd_sharded_sort_indices: no communicator / out_start");
    ARROW_ASSIGN_OR_RAISE(auto v, arrow::ImportDeviceArray(values, values_schema));
    if (stage_ms != nullptr) std::fill(stage_ms, stage_ms + 4, 0.0);
    std::shared_ptr<ArrayData> idx;
    ARROW_RETURN_NOT_OK(ShardedSortIndices(*static_cast<ShardedComm*>(comm), *v->data(), descending != 0, nulls_first != 0,
                                           splitter_bits, &idx, out_start, stage_ms));
    ARROW_RETURN_NOT_OK(arrow::ExportType(*idx->type, out_schema));
    return arrow::ExportDeviceArray(*arrow::MakeArray(idx), nullptr, out_indices);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// select_k_unstable by a threshold (no sort of the column): the row floor; how often it ran
void arrow_amd_plugin_set_select_k_min_rows(int64_t rows) { g_select_k_min_rows.store(rows < 0 ? 0 : rows); }
int64_t arrow_amd_plugin_select_k_threshold_runs(void) { return g_select_k_threshold_runs.load(); }
// the sharded group-by's range-partitioned state (on / off; the row floor of a single rank), how often it ran / was declined together
void arrow_amd_plugin_set_sharded_range_state(int on, int64_t min_rows) {
  g_sharded_range_state.store(on != 0);
  g_sharded_range_min_rows.store(min_rows < 0 ? 0 : min_rows);
}
int64_t arrow_amd_plugin_sharded_range_runs(void) { return g_sharded_range_runs.load(); }
int64_t arrow_amd_plugin_sharded_range_declined(void) { return g_sharded_range_declined.load(); }
// the sharded sort's records form (on / off), its sample (shift 0 .. 8, 0 = exact; the longest shard's row floor), how often it ran
void arrow_amd_plugin_set_sharded_sort_records(int on) { g_sharded_sort_records.store(on != 0); }
void arrow_amd_plugin_set_sharded_sort_sample(int shift, int64_t min_rows) {
  g_sharded_sort_sample_shift.store(shift < 0 ? 0 : (shift > 8 ? 8 : shift));
  g_sharded_sort_sample_min_rows.store(min_rows < 0 ? 0 : min_rows);
}
int64_t arrow_amd_plugin_sharded_sort_records_runs(void) { return g_sharded_sort_records_runs.load(); }
void arrow_amd_plugin_pool_stats(int64_t* cached_bytes, int64_t* hits, int64_t* misses) {
  DevicePool::Get().Stats(cached_bytes, hits, misses);
}
// Returns every cached block to the driver.
void arrow_amd_plugin_pool_trim(void) { DevicePool::Get().Trim(); }
// One column chunk of a Parquet file decoded into a device-resident array (C Device Data interface):
// the reference's PageReader for headers + decompression, the C-ABI kernels for everything per value.
int arrow_amd_parquet_read_column(const char* path, int row_group, int column, struct ArrowDeviceArray* out,
                                  struct ArrowSchema* out_schema) {
  auto run = [&]() -> Status {
    try {
      ARROW_ASSIGN_OR_RAISE(auto data, ParquetChunkToDevice(path, row_group, column));
      ARROW_RETURN_NOT_OK(arrow::ExportType(*data->type, out_schema));
      return arrow::ExportDeviceArray(*arrow::MakeArray(data), nullptr, out);
    } catch (const parquet::ParquetException& e) {
      return Status::IOError("Parquet: ", e.what());
    }
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// One TOP-LEVEL FIELD of a row group (flat column, list chain or struct of primitives) as a device-resident array: what the
// reference's FileReader::ReadRowGroup assembles per schema field (parquet/arrow/reader.cc).
int arrow_amd_parquet_read_field(const char* path, int row_group, int field, struct ArrowDeviceArray* out,
                                 struct ArrowSchema* out_schema) {
  auto run = [&]() -> Status {
    try {
      ARROW_ASSIGN_OR_RAISE(auto data, ParquetFieldToDevice(path, row_group, field));
      ARROW_RETURN_NOT_OK(arrow::ExportType(*data->type, out_schema));
      return arrow::ExportDeviceArray(*arrow::MakeArray(data), nullptr, out);
    } catch (const parquet::ParquetException& e) {
      return Status::IOError("Parquet: ", e.what());
    }
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// Several column chunks of one row group at once.  A column chunk is decoded by one host thread and (for its Snappy
// pages) one wave per page — with ~1000 pages that is one wave per SIMD and nothing to hide its latency behind — so
// the chunks of a row group are given to a small pool of worker threads that live as long as the library: every
// worker keeps its own stream, device scratch and page-locked / host buffers from call to call (what makes a chunk
// cheap), and their kernels and copies overlap on the device.  Results in column order; the first error
// wins.  arrow_amd_parquet_read_column stays the one-chunk form.
namespace {
class ChunkWorkers {
 public:
  static ChunkWorkers& Get() {
#ifdef ARX_EMULATED_HIP_RUNTIME   // (tests/emu: the emulator runs one kernel at a time)
    static ChunkWorkers* pool = new ChunkWorkers(1);
#else
    static ChunkWorkers* pool = new ChunkWorkers(4);   // (never destroyed: worker threads must not outlive their statics)
#endif
    return *pool;
  }
  // runs fn(i) for i in [0, n) on the workers; returns when all are done
  void Run(int n, const std::function<void(int)>& fn) {
    std::unique_lock<std::mutex> lock(mu_);
    idle_.wait(lock, [&] { return !busy_; });          // one Run at a time
    busy_ = true;
    fn_ = &fn;
    next_ = 0;
    count_ = n;
    left_ = n;
    work_.notify_all();
    done_.wait(lock, [&] { return left_ == 0; });
    fn_ = nullptr;
    busy_ = false;
    idle_.notify_one();
  }

 private:
  explicit ChunkWorkers(int k) {
    for (int i = 0; i < k; ++i) std::thread([this] { Loop(); }).detach();
  }
  void Loop() {
    std::unique_lock<std::mutex> lock(mu_);
    for (;;) {
      work_.wait(lock, [&] { return fn_ != nullptr && next_ < count_; });
      const int i = next_++;
      const std::function<void(int)>* fn = fn_;
      lock.unlock();
      (*fn)(i);
      lock.lock();
      if (--left_ == 0) done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable work_, done_, idle_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, count_ = 0, left_ = 0;
  bool busy_ = false;
};
}  // namespace

int arrow_amd_parquet_read_columns(const char* path, int row_group, const int* columns, int num_columns,
                                   struct ArrowDeviceArray* outs, struct ArrowSchema* out_schemas) {
  if (path == nullptr || columns == nullptr || num_columns < 0 || (num_columns > 0 && (outs == nullptr || out_schemas == nullptr))) {
    t_error = "arrow_amd_parquet_read_columns: NULL argument";
    return -1;
  }
  std::vector<arrow::Result<std::shared_ptr<ArrayData>>> results(static_cast<size_t>(num_columns),
                                                                  arrow::Result<std::shared_ptr<ArrayData>>(Status::UnknownError("not run")));
  const std::string file(path);
  int device = 0;
  (void)hipGetDevice(&device);   // the workers decode on the caller's device
  ChunkWorkers::Get().Run(num_columns, [&](int i) {
    (void)hipSetDevice(device);
    try {
      results[i] = ParquetChunkToDevice(file, row_group, columns[i]);
    } catch (const parquet::ParquetException& e) {
      results[i] = Status::IOError("Parquet: ", e.what());
    } catch (const std::exception& e) {
      results[i] = Status::UnknownError(e.what());
    }
  });
  for (int i = 0; i < num_columns; ++i) {
    if (!results[i].ok()) {
      t_error = results[i].status().ToString();
      return -1;
    }
  }
  int exported = 0;
  Status st;
  for (; exported < num_columns && st.ok(); ++exported) {
    const std::shared_ptr<ArrayData>& data = *results[exported];
    st = arrow::ExportType(*data->type, &out_schemas[exported]);
    if (st.ok()) {
      st = arrow::ExportDeviceArray(*arrow::MakeArray(data), nullptr, &outs[exported]);
      if (!st.ok() && out_schemas[exported].release != nullptr) out_schemas[exported].release(&out_schemas[exported]);
    }
  }
  if (!st.ok()) {   // give back what was already exported
    for (int i = 0; i + 1 < exported; ++i) {
      if (outs[i].array.release != nullptr) outs[i].array.release(&outs[i].array);
      if (out_schemas[i].release != nullptr) out_schemas[i].release(&out_schemas[i]);
    }
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// Inputs shorter than this stay on the stock CPU kernels (PCIe staging does not pay).
// 1 = OPT-IN: unmodified Acero plans (table_source / aggregate / order_by by their stock names) over device-resident tables
// land on the plugin's nodes; 0 = what arrow_amd_register() leaves: the guard below + stock `table_source` nodes over
// device-resident tables deliver whole chunks instead of 32Ki-row morsels (round 6); -1 = the guard alone (the stock
// source's morsels).  Returns -1 with arrow_amd_plugin_last_error() when the default registry of this Arrow build was
// not recognised (nothing is changed then).  plugin/acero_override.inc
// The guard of plugin/acero_override.inc: 1 = installed by arrow_amd_register() (a keyed aggregation over device-resident key
// columns is built as aggregate_rocm or refused with a Status), 0 = the registry's layout was not recognised.  what = 1:
// aggregations it turned into aggregate_rocm so far; what = 2: aggregations it refused.
int64_t arrow_amd_plugin_acero_guard(int what) {
  return what == 1 ? g_guard_takeovers.load() : what == 2 ? g_guard_refusals.load() : (g_acero_guard_installed.load() ? 1 : 0);
}
int arrow_amd_override_acero_factories(int on) {
  const Status st = OverrideAceroFactories(on);
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
void arrow_amd_plugin_set_min_rows(int64_t n) { g_min_rows.store(n); }
// aggregate_rocm's var-width keys: bits of the strings' hash (64; 0 = the exact 12-byte chunk columns only; a few bits
// force collisions, for tests) and how many batches were regrouped after a verified collision
void arrow_amd_plugin_set_string_key_hash_bits(int64_t bits) { g_string_key_hash_bits.store(bits < 0 ? 0 : (bits > 64 ? 64 : bits)); }
int64_t arrow_amd_plugin_string_key_hash_collisions(void) { return g_string_key_hash_collisions.load(); }
// Device-resident filters of at most n rows use one synchronisation instead of three (0 = off, the default).
void arrow_amd_plugin_set_filter_morsel_rows(int64_t n) { g_filter_morsel_rows.store(n); }
// Parquet: 1 (default) = Snappy chunks of fixed-width columns are read raw and their PLAIN value pages decompressed
// on the device; 0 = every page is decompressed by the reference's PageReader on the host
void arrow_amd_plugin_set_parquet_device_snappy(int on) { g_parquet_device_snappy.store(on != 0); }
// A/B switch: definition levels of all-V2 chunks walked on the device (default) or by the host's run scanner.
void arrow_amd_plugin_set_parquet_device_levels(int on) { g_parquet_device_levels.store(on != 0); }
// A/B switch: gather the compressed pages in page-locked host memory (default) or in a pageable vector.
void arrow_amd_plugin_set_parquet_pinned_staging(int on) { g_parquet_pinned_staging.store(on != 0); }
// threads that share the read of one column chunk, each reading at least min_part_bytes (default 4 x 8 MB; 1 = a single ReadAt)
void arrow_amd_plugin_set_parquet_read_threads(int n, int64_t min_part_bytes) {
  g_parquet_read_threads.store(n < 1 ? 1 : (n > 16 ? 16 : n));
  g_parquet_read_part_bytes.store(min_part_bytes < 4096 ? 4096 : min_part_bytes);
}
// pages decompressed on the device so far
int64_t arrow_amd_plugin_parquet_device_snappy_pages(void) { return g_parquet_device_snappy_pages.load(); }
void arrow_amd_plugin_set_parquet_device_gzip(int on) { g_parquet_device_gzip.store(on != 0); }
int64_t arrow_amd_plugin_parquet_device_gzip_pages(void) { return g_parquet_device_gzip_pages.load(); }
// device-route pages that had to be copied into the staging block (0: the page reader hands out slices of the chunk)
int64_t arrow_amd_plugin_parquet_copied_pages(void) { return g_parquet_copied_pages.load(); }
// aggregate_rocm: rows of pending device batches that trigger a copy into the staging columns; copies so far
void arrow_amd_plugin_set_aggregate_stage_nulls(int on) { g_aggregate_stage_nulls.store(on != 0); }
void arrow_amd_plugin_set_aggregate_flush_rows(int64_t rows) { g_aggregate_flush_rows.store(rows < 1 ? 1 : rows); }
int64_t arrow_amd_plugin_aggregate_flushes(void) { return g_aggregate_flushes.load(); }
// aggregate_rocm consumes a device batch of at least this many rows where it lies (no staging copy)
void arrow_amd_plugin_set_aggregate_direct_rows(int64_t rows) { g_aggregate_direct_rows.store(rows < 1 ? 1 : rows); }
int64_t arrow_amd_plugin_aggregate_direct_batches(void) { return g_aggregate_direct_batches.load(); }
// aggregate_rocm's range-partitioned state for a plan that is one run of device slices (default on); its row floor; how many
// plans went through it / how many it declined to the table (a wide range, a key the sample missed, a hot key)
void arrow_amd_plugin_set_aggregate_range_state(int on) { g_aggregate_range_state.store(on != 0); }
void arrow_amd_plugin_set_aggregate_range_min_rows(int64_t rows) { g_aggregate_range_min_rows.store(rows < 1 ? 1 : rows); }
int64_t arrow_amd_plugin_aggregate_range_plans(void) { return g_aggregate_range_plans.load(); }
int64_t arrow_amd_plugin_aggregate_range_declined(void) { return g_aggregate_range_declined.load(); }
// aggregate_rocm keeps the result of a plan over device-resident rows in HBM (default off: host arrays, as GroupByNode)
void arrow_amd_plugin_set_aggregate_device_output(int on) { g_aggregate_device_output.store(on != 0); }
// column-sized result copies by a kernel (arx_buffer_copy) instead of the copy engines (default on; A/B knob)
void arrow_amd_plugin_set_results_kernel_copy(int on) { g_results_kernel_copy.store(on != 0); }
// result columns >= 1 MB into pooled page-locked host buffers (default on)
void arrow_amd_plugin_set_pinned_results(int on) { g_pinned_results.store(on != 0); }
// coalesce_rocm: rows gathered into one batch (at least); morsels it has joined so far
void arrow_amd_plugin_set_coalesce_rows(int64_t rows) { g_coalesce_rows.store(rows < 1 ? 1 : rows); }
int64_t arrow_amd_plugin_coalesced_batches(void) { return g_coalesced_batches.load(); }
// table_source_rocm: rows per batch when TableSourceNodeOptions::max_batch_size is the default
void arrow_amd_plugin_set_table_source_rows(int64_t rows) { g_table_source_rows.store(rows < 1 ? 1 : rows); }
// The same threshold for the element-wise kernels (greater, cast); default: never stage them.
void arrow_amd_plugin_set_min_rows_streaming(int64_t n) { g_min_rows_streaming.store(n); }

}  // extern "C"
