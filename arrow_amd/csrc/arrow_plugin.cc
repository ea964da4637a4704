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
// This round the arrays the Arrow API hands over live in HOST memory (ArraySpan::buffers[i].data,
// cpp/src/arrow/array/data.h:525-553), so every call stages through HBM over PCIe: correct and
// drop-in, but bandwidth-bound by the link, not by HBM.  Device-resident ExecBatches
// (a kROCM arrow::Device/MemoryManager/Buffer, cpp/src/arrow/device.h:43-280) are row (f1) of
// SURVEY.md section 8 and are what bench.py measures through arrow_amd.compute.
#include <arrow/acero/exec_plan.h>
#include <arrow/acero/options.h>
#include <arrow/api.h>
#include <arrow/c/abi.h>
#include <arrow/c/bridge.h>
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

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
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

std::atomic<int64_t> g_gpu_calls{0};
std::atomic<int64_t> g_stock_calls{0};
std::atomic<int64_t> g_min_rows{1 << 16};
// Element-wise kernels (greater, cast) on HOST arrays move more bytes over PCIe than the CPU needs
// time to compute them (100M rows, one MI355X box: greater 38 ms staged vs 24 ms stock; cast 27 vs
// 33 ms; filter 17 vs 116 ms; sort 46 ms vs 14.8 s — scripts/exp_plugin_host_staging.py), so by
// default they stay on Arrow's stock kernels; device-resident arrays always run on the GPU.
std::atomic<int64_t> g_min_rows_streaming{INT64_MAX};

// per-function call counters: which exec actually ran (the GPU tests assert on these so that a
// silent route through the stock CPU kernel is a test failure, not a pass)
enum Fn { kFnFilter = 0, kFnTake, kFnGreater, kFnSort, kFnCast, kFnHashSum, kFnAdd, kFnBoolean, kFnCompare, kNumFn };
const char* const kFnNames[kNumFn] = {"array_filter", "array_take", "greater", "array_sort_indices",
                                      "cast", "hash_sum", "add", "boolean", "compare"};
std::atomic<int64_t> g_fn_gpu[kNumFn];
std::atomic<int64_t> g_fn_stock[kNumFn];
void CountGpu(Fn f) {
  g_gpu_calls.fetch_add(1, std::memory_order_relaxed);
  g_fn_gpu[f].fetch_add(1, std::memory_order_relaxed);
}
void CountStock(Fn f) {
  g_stock_calls.fetch_add(1, std::memory_order_relaxed);
  g_fn_stock[f].fetch_add(1, std::memory_order_relaxed);
}
thread_local std::string t_error;

Status FromArx(int rc) {
  if (rc == ARX_OK) return Status::OK();
  const std::string msg = arx_last_error();
  switch (rc) {
    case ARX_INVALID: return Status::Invalid(msg);
    case ARX_INDEX_ERROR: return Status::IndexError(msg);
    case ARX_NOT_IMPLEMENTED: return Status::NotImplemented(msg);
    case ARX_OUT_OF_MEMORY: return Status::OutOfMemory(msg);
    default: return Status::UnknownError(msg);
  }
}

#define HIP_RETURN_NOT_OK(call)                                                        \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return Status::IOError("HIP: ", hipGetErrorString(e__), " in " #call);           \
  } while (0)

// ---------------------------------------------------------------- per-thread device scratch
// Kernel::exec may be called concurrently from Acero's thread pool with the same const Kernel*
// (SURVEY.md 8b): every host thread owns a stream and a handful of growable HBM staging slots.
class DeviceScratch {
 public:
  ~DeviceScratch() {
    for (auto& s : slots_) {
      if (s.ptr) (void)hipFree(s.ptr);
    }
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  Status Stream(hipStream_t* out) {
    if (!stream_) HIP_RETURN_NOT_OK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    *out = stream_;
    return Status::OK();
  }
  Status Get(int slot, size_t bytes, void** out) {
    if (slots_.size() <= static_cast<size_t>(slot)) slots_.resize(slot + 1);
    Slot& s = slots_[slot];
    bytes = (bytes + 255) & ~size_t(255);
    if (s.size < bytes) {
      if (s.ptr) HIP_RETURN_NOT_OK(hipFree(s.ptr));
      s.ptr = nullptr;
      s.size = 0;
      const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
      HIP_RETURN_NOT_OK(hipMalloc(&s.ptr, want));
      s.size = want;
    }
    *out = s.ptr;
    return Status::OK();
  }

 private:
  struct Slot { void* ptr = nullptr; size_t size = 0; };
  std::vector<Slot> slots_;
  hipStream_t stream_ = nullptr;
};
thread_local DeviceScratch t_scratch;

enum Slot { kValues = 0, kValidity, kArg2, kArg2Validity, kOutData, kOutValidity, kWs, kCounter, kBinWs, kFlag };

// Upload the logical range of a fixed-width (or boolean) ArraySpan.  The device copy keeps the
// sub-byte part of the offset (offset % 8) so that one logical offset addresses both buffers.
Status Upload(const ArraySpan& a, int byte_width_or_0_for_bool, int data_slot, int validity_slot,
              hipStream_t st, ArxSpan* out) {
  const int64_t o8 = a.offset % 8;
  const int64_t first = a.offset - o8;
  out->offset = o8;
  out->length = a.length;
  out->null_count = a.null_count;
  out->validity = nullptr;
  out->data = nullptr;
  const int64_t span_elems = o8 + a.length;
  if (a.buffers[1].data != nullptr && a.length > 0) {
    const uint8_t* src;
    size_t bytes;
    if (byte_width_or_0_for_bool == 0) {
      src = a.buffers[1].data + first / 8;
      bytes = static_cast<size_t>(arrow::bit_util::BytesForBits(span_elems));
    } else {
      src = a.buffers[1].data + first * byte_width_or_0_for_bool;
      bytes = static_cast<size_t>(span_elems) * byte_width_or_0_for_bool;
    }
    void* d = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(data_slot, bytes + 16, &d));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
    out->data = d;
  }
  if (a.buffers[0].data != nullptr && a.null_count != 0 && a.length > 0) {
    const size_t bytes = static_cast<size_t>(arrow::bit_util::BytesForBits(span_elems));
    void* d = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(validity_slot, bytes + 16, &d));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(d, a.buffers[0].data + first / 8, bytes, hipMemcpyHostToDevice, st));
    out->validity = d;
  } else if (a.null_count != 0 && a.buffers[0].data == nullptr) {
    out->null_count = 0;  // no bitmap: all valid
  }
  return Status::OK();
}

bool IsHost(const ArraySpan& a) {
  for (int i = 0; i < 2; ++i) {
    if (a.buffers[i].owner != nullptr && *a.buffers[i].owner != nullptr &&
        !(*a.buffers[i].owner)->is_cpu()) {
      return false;
    }
  }
  return true;
}

int FixedByteWidth(const arrow::DataType& t) {
  switch (t.id()) {
    case Type::INT8: case Type::UINT8: return 1;
    case Type::INT16: case Type::UINT16: case Type::HALF_FLOAT: return 2;
    case Type::INT32: case Type::UINT32: case Type::FLOAT: case Type::DATE32: case Type::TIME32:
    case Type::INTERVAL_MONTHS: case Type::DECIMAL32: return 4;
    case Type::INT64: case Type::UINT64: case Type::DOUBLE: case Type::DATE64: case Type::TIME64:
    case Type::TIMESTAMP: case Type::DURATION: case Type::INTERVAL_DAY_TIME: case Type::DECIMAL64: return 8;
    case Type::DECIMAL128: case Type::INTERVAL_MONTH_DAY_NANO: return 16;
    case Type::FIXED_SIZE_BINARY: {
      const int w = static_cast<const arrow::FixedSizeBinaryType&>(t).byte_width();
      return (w == 1 || w == 2 || w == 4 || w == 8 || w == 16) ? w : 0;  // other widths: stock kernel
    }
    default: return 0;
  }
}

// ---------------------------------------------------------------- device-resident arrays
// A minimal kROCM arrow::Device / MemoryManager / Buffer (interfaces: cpp/src/arrow/device.h:43-280,
// buffer.h:52-; pattern: cpp/src/arrow/gpu/cuda_memory.{h,cc}) so that Arrow arrays can LIVE in
// HBM: ArraySpan::buffers[i].data is null for such buffers (buffer.h:221-226) and the kernels
// below read the device address from buffers[i].owner->address() — no staging, outputs are
// RocmBuffers too.  The C Device Data interface (c/abi.h ARROW_DEVICE_ROCM, c/bridge.h) is how
// arrays enter/leave: RegisterDeviceMapper(kROCM) lets ImportDeviceArray (and therefore
// pyarrow.Array._import_from_c_device) build arrays on this memory manager.
class RocmMemoryManager;

class RocmDevice : public arrow::Device {
 public:
  explicit RocmDevice(int id) : arrow::Device(/*is_cpu=*/false), id_(id) {}
  const char* type_name() const override { return "arrow_amd::RocmDevice"; }
  std::string ToString() const override { return "RocmDevice(gfx950, device_id=" + std::to_string(id_) + ")"; }
  bool Equals(const arrow::Device& other) const override {
    return other.device_type() == device_type() && other.device_id() == id_;
  }
  int64_t device_id() const override { return id_; }
  arrow::DeviceAllocationType device_type() const override { return arrow::DeviceAllocationType::kROCM; }
  std::shared_ptr<arrow::MemoryManager> default_memory_manager() override;

 private:
  int id_;
};

// Caching device allocator behind RocmMemoryManager: hipMalloc / hipFree cost tens of microseconds
// each and serialise the device, which dominates Acero-sized (32K-row) batches — a filter
// allocates two buffers per column per batch.  Freed blocks are kept on per-size-class free lists
// (power-of-two classes below 1 MiB, 8 classes per octave above: <= 12.5 % slack) up to a byte
// limit (ARROW_AMD_POOL_LIMIT_MB, default 32768).  Reuse is safe without stream ordering because
// every shim synchronises its stream before it returns, i.e. before Arrow can drop a buffer.
class DevicePool {
 public:
  static DevicePool& Get() {
    static DevicePool* pool = new DevicePool();  // leaked on purpose: buffers may outlive static destruction
    return *pool;
  }
  static size_t SizeClass(size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    size_t p2 = 256;
    while (p2 < bytes) p2 <<= 1;
    if (p2 <= (size_t(1) << 20)) return p2;
    const size_t step = p2 >> 4;  // (p2/2, p2] in 8 steps of p2/16
    return (bytes + step - 1) / step * step;
  }
  Status Allocate(size_t bytes, void** out, size_t* cap) {
    const size_t c = SizeClass(bytes);
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = free_.find(c);
      if (it != free_.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        cached_bytes_ -= c;
        ++hits_;
        *cap = c;
        return Status::OK();
      }
      ++misses_;
    }
    hipError_t e = hipMalloc(out, c);
    if (e != hipSuccess) {
      Trim();  // give the cached blocks back and retry once
      e = hipMalloc(out, c);
    }
    if (e != hipSuccess) return Status::OutOfMemory("hipMalloc of ", c, " bytes: ", hipGetErrorString(e));
    *cap = c;
    return Status::OK();
  }
  void Release(void* p, size_t cap) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (cached_bytes_ + cap <= limit_) {
        free_[cap].push_back(p);
        cached_bytes_ += cap;
        return;
      }
    }
    (void)hipFree(p);
  }
  void Trim() {
    std::unordered_map<size_t, std::vector<void*>> drop;
    {
      std::lock_guard<std::mutex> lock(mu_);
      drop.swap(free_);
      cached_bytes_ = 0;
    }
    for (auto& kv : drop) {
      for (void* p : kv.second) (void)hipFree(p);
    }
  }
  void Stats(int64_t* cached, int64_t* hits, int64_t* misses) {
    std::lock_guard<std::mutex> lock(mu_);
    *cached = static_cast<int64_t>(cached_bytes_);
    *hits = hits_;
    *misses = misses_;
  }

 private:
  DevicePool() {
    const char* env = std::getenv("ARROW_AMD_POOL_LIMIT_MB");
    limit_ = (env != nullptr ? static_cast<size_t>(std::strtoull(env, nullptr, 10)) : size_t(32768)) << 20;
  }
  std::mutex mu_;
  std::unordered_map<size_t, std::vector<void*>> free_;
  size_t cached_bytes_ = 0, limit_ = 0;
  int64_t hits_ = 0, misses_ = 0;
};

// an owned device allocation (returned to the pool on destruction)
class RocmBuffer : public arrow::MutableBuffer {
 public:
  RocmBuffer(uint8_t* ptr, int64_t size, size_t capacity, std::shared_ptr<arrow::MemoryManager> mm)
      : arrow::MutableBuffer(ptr, size, std::move(mm)), ptr_(ptr), capacity_(capacity) {}
  ~RocmBuffer() override {
    if (ptr_) DevicePool::Get().Release(ptr_, capacity_);
  }

 private:
  uint8_t* ptr_;
  size_t capacity_;
};

class RocmMemoryManager : public arrow::MemoryManager {
 public:
  explicit RocmMemoryManager(const std::shared_ptr<arrow::Device>& device) : arrow::MemoryManager(device) {}

  arrow::Result<std::shared_ptr<arrow::io::RandomAccessFile>> GetBufferReader(std::shared_ptr<Buffer>) override {
    return Status::NotImplemented("RocmMemoryManager::GetBufferReader");
  }
  arrow::Result<std::shared_ptr<arrow::io::OutputStream>> GetBufferWriter(std::shared_ptr<Buffer>) override {
    return Status::NotImplemented("RocmMemoryManager::GetBufferWriter");
  }
  arrow::Result<std::unique_ptr<Buffer>> AllocateBuffer(int64_t size) override {
    void* p = nullptr;
    size_t cap = 0;
    // padded like Arrow's pools (64 bytes) so that whole-word bitmap stores stay inside
    const size_t bytes = (static_cast<size_t>(std::max<int64_t>(size, 1)) + 63) & ~size_t(63);
    ARROW_RETURN_NOT_OK(DevicePool::Get().Allocate(bytes, &p, &cap));
    return std::unique_ptr<Buffer>(new RocmBuffer(static_cast<uint8_t*>(p), size, cap, shared_from_this()));
  }

 protected:
  arrow::Result<std::shared_ptr<Buffer>> CopyBufferFrom(const std::shared_ptr<Buffer>& buf,
                                                        const std::shared_ptr<arrow::MemoryManager>& from) override {
    ARROW_ASSIGN_OR_RAISE(auto out, CopyNonOwnedFrom(*buf, from));
    return std::shared_ptr<Buffer>(std::move(out));
  }
  arrow::Result<std::shared_ptr<Buffer>> CopyBufferTo(const std::shared_ptr<Buffer>& buf,
                                                      const std::shared_ptr<arrow::MemoryManager>& to) override {
    ARROW_ASSIGN_OR_RAISE(auto out, CopyNonOwnedTo(*buf, to));
    return std::shared_ptr<Buffer>(std::move(out));
  }
  arrow::Result<std::unique_ptr<Buffer>> CopyNonOwnedFrom(const Buffer& buf,
                                                          const std::shared_ptr<arrow::MemoryManager>& from) override {
    if (!from->is_cpu()) return nullptr;  // unsupported pair: let Arrow try the other direction
    ARROW_ASSIGN_OR_RAISE(auto out, AllocateBuffer(buf.size()));
    if (buf.size() > 0) {
      HIP_RETURN_NOT_OK(hipMemcpy(reinterpret_cast<void*>(out->mutable_address()), buf.data(),
                                  static_cast<size_t>(buf.size()), hipMemcpyHostToDevice));
    }
    return out;
  }
  arrow::Result<std::unique_ptr<Buffer>> CopyNonOwnedTo(const Buffer& buf,
                                                        const std::shared_ptr<arrow::MemoryManager>& to) override {
    if (!to->is_cpu()) return nullptr;
    ARROW_ASSIGN_OR_RAISE(auto out, to->AllocateBuffer(buf.size()));
    if (buf.size() > 0) {
      HIP_RETURN_NOT_OK(hipMemcpy(out->mutable_data(), reinterpret_cast<const void*>(buf.address()),
                                  static_cast<size_t>(buf.size()), hipMemcpyDeviceToHost));
    }
    return out;
  }
};

std::shared_ptr<arrow::MemoryManager> RocmDevice::default_memory_manager() {
  static std::mutex mu;
  static std::vector<std::shared_ptr<arrow::MemoryManager>> cache;
  std::lock_guard<std::mutex> lock(mu);
  if (cache.size() <= static_cast<size_t>(id_)) cache.resize(id_ + 1);
  if (!cache[id_]) cache[id_] = std::make_shared<RocmMemoryManager>(shared_from_this());
  return cache[id_];
}

arrow::Result<std::shared_ptr<arrow::MemoryManager>> RocmMemoryManagerFor(int64_t device_id) {
  static std::mutex mu;
  static std::vector<std::shared_ptr<RocmDevice>> devices;
  std::lock_guard<std::mutex> lock(mu);
  if (device_id < 0) device_id = 0;
  if (devices.size() <= static_cast<size_t>(device_id)) devices.resize(device_id + 1);
  if (!devices[device_id]) devices[device_id] = std::make_shared<RocmDevice>(static_cast<int>(device_id));
  return devices[device_id]->default_memory_manager();
}

// true if any buffer of the span lives on a kROCM device
bool OnRocm(const ArraySpan& a) {
  for (int i = 0; i < 2; ++i) {
    if (a.buffers[i].owner != nullptr && *a.buffers[i].owner != nullptr &&
        (*a.buffers[i].owner)->device_type() == arrow::DeviceAllocationType::kROCM) {
      return true;
    }
  }
  return false;
}

bool DataOnRocm(const ArrayData& a) {
  for (const auto& b : a.buffers) {
    if (b != nullptr && b->device_type() == arrow::DeviceAllocationType::kROCM) return true;
  }
  return false;
}

// ArxSpan over device-resident buffers: addresses come from the owning Buffer, not from .data
Status DeviceSpan(const ArraySpan& a, ArxSpan* out) {
  out->offset = a.offset;
  out->length = a.length;
  out->validity = nullptr;
  out->data = nullptr;
  for (int i = 0; i < 2; ++i) {
    const auto* owner = a.buffers[i].owner;
    if (owner == nullptr || *owner == nullptr) continue;
    if ((*owner)->device_type() != arrow::DeviceAllocationType::kROCM) {
      return Status::Invalid("arrow_amd: mixed host / device buffers in one array");
    }
    (i == 0 ? out->validity : out->data) = reinterpret_cast<const void*>((*owner)->address());
  }
  // The executor zeroes ArraySpan::null_count when buffers[0].data is null — which it is for
  // every non-CPU buffer (Buffer::data(), buffer.h:221-226) — so the presence of the validity
  // BUFFER is the only signal left: report "unknown" and let the kernels read the bitmap.
  out->null_count = out->validity == nullptr ? 0 : (a.null_count > 0 ? a.null_count : arrow::kUnknownNullCount);
  return Status::OK();
}

arrow::Result<std::shared_ptr<Buffer>> AllocDevice(int64_t bytes) {
  ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
  ARROW_ASSIGN_OR_RAISE(auto buf, mm->AllocateBuffer(bytes));
  return std::shared_ptr<Buffer>(std::move(buf));
}

// exact null count of a device bitmap (a device array must never need a CPU popcount later)
arrow::Result<int64_t> DeviceNullCount(const Buffer& bitmap, int64_t length, hipStream_t st) {
  if (length == 0) return 0;
  void* ws = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kCounter, 64, &ws));
  int64_t set_bits = 0;
  ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_popcount(reinterpret_cast<const void*>(bitmap.address()), 0, length, ws,
                                                  64, &set_bits, st)));
  return length - set_bits;
}

// ---------------------------------------------------------------- state = stock state + options
template <typename Options>
struct ShimState : public cp::KernelState {
  std::unique_ptr<cp::KernelState> stock;
  Options options;
};

struct StockKernel {
  cp::KernelInit init;
  cp::ArrayKernelExec exec = nullptr;
  cp::VectorKernel::ChunkedExec exec_chunked = nullptr;
};

// run the stock exec with the stock state installed
Status RunStock(Fn fn, const StockKernel& k, cp::KernelState* stock_state, cp::KernelContext* ctx,
                const cp::ExecSpan& batch, cp::ExecResult* out) {
  CountStock(fn);
  cp::KernelState* mine = ctx->state();
  ctx->SetState(stock_state);
  Status st = k.exec(ctx, batch, out);
  ctx->SetState(mine);
  return st;
}

// ---------------------------------------------------------------- filter
StockKernel g_stock_filter;

arrow::Result<std::unique_ptr<cp::KernelState>> FilterInit(cp::KernelContext* ctx,
                                                           const cp::KernelInitArgs& args) {
  auto state = std::make_unique<ShimState<cp::FilterOptions>>();
  if (g_stock_filter.init) {
    ARROW_ASSIGN_OR_RAISE(state->stock, g_stock_filter.init(ctx, args));
  }
  if (args.options != nullptr) {
    state->options = *static_cast<const cp::FilterOptions*>(args.options);
  }
  return state;
}

// PrimitiveFilterExec (vector_selection_filter_internal.cc:445-510) with the loop on the GPU.
Status FilterExec(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  auto* state = static_cast<ShimState<cp::FilterOptions>*>(ctx->state());
  const ArraySpan& values = batch[0].array;
  const ArraySpan& filter = batch[1].array;
  const int w = FixedByteWidth(*values.type);
  const int null_sel = state->options.null_selection_behavior == cp::FilterOptions::EMIT_NULL
                           ? ARX_FILTER_EMIT_NULL : ARX_FILTER_DROP;
  const bool on_device = OnRocm(values) || OnRocm(filter);
  if (on_device) {
    // device-resident ExecBatch: no staging, the output stays in HBM
    if (w == 0 || filter.type->id() != Type::BOOL) {
      return Status::NotImplemented("arrow_amd: filter of ", values.type->ToString(),
                                    " on device-resident arrays");
    }
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ArxSpan dv{}, dm{};
    ARROW_RETURN_NOT_OK(DeviceSpan(values, &dv));
    ARROW_RETURN_NOT_OK(DeviceSpan(filter, &dm));
    const size_t ws_bytes = arx_filter_workspace_bytes(filter.length);
    void* ws = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes, &ws));
    int64_t out_len = 0;
    ARROW_RETURN_NOT_OK(FromArx(arx_filter_count(&dm, null_sel, ws, ws_bytes, &out_len, st)));
    ArrayData* out_arr = out->array_data().get();
    const bool allocate_validity = dv.null_count != 0 || dm.null_count != 0;
    out_arr->length = out_len;
    out_arr->buffers.resize(2);
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(out_len * w));
    out_arr->buffers[0] = nullptr;
    void* d_valid = nullptr;
    if (allocate_validity) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(((out_len + 63) / 64) * 8));
      d_valid = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
    }
    ARROW_RETURN_NOT_OK(FromArx(arx_filter_exec(&dv, w, &dm, null_sel, ws, out_len,
                                                reinterpret_cast<void*>(out_arr->buffers[1]->mutable_address()),
                                                d_valid, st)));
    out_arr->null_count = 0;
    if (allocate_validity) {
      ARROW_ASSIGN_OR_RAISE(out_arr->null_count, DeviceNullCount(*out_arr->buffers[0], out_len, st));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnFilter);
    return Status::OK();
  }
  if (w == 0 || filter.type->id() != Type::BOOL || values.length < g_min_rows.load() ||
      !IsHost(values) || !IsHost(filter)) {
    return RunStock(kFnFilter, g_stock_filter, state->stock.get(), ctx, batch, out);
  }
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxSpan dv{}, dm{};
  ARROW_RETURN_NOT_OK(Upload(values, w, kValues, kValidity, st, &dv));
  ARROW_RETURN_NOT_OK(Upload(filter, 0, kArg2, kArg2Validity, st, &dm));
  const size_t ws_bytes = arx_filter_workspace_bytes(filter.length);
  void* ws = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes, &ws));
  int64_t out_len = 0;
  ARROW_RETURN_NOT_OK(FromArx(arx_filter_count(&dm, null_sel, ws, ws_bytes, &out_len, st)));

  ArrayData* out_arr = out->array_data().get();
  const bool filter_null_count_is_zero = filter.null_count == 0;
  if (values.null_count == 0 && (null_sel == ARX_FILTER_DROP || filter_null_count_is_zero)) {
    out_arr->null_count = 0;
  } else {
    out_arr->null_count = arrow::kUnknownNullCount;
  }
  const bool allocate_validity = values.null_count != 0 || !filter_null_count_is_zero;
  out_arr->length = out_len;
  out_arr->buffers.resize(2);
  ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], ctx->Allocate(out_len * w));
  void* d_out = nullptr;
  void* d_out_valid = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, static_cast<size_t>(out_len) * w + 16, &d_out));
  const size_t vbytes = static_cast<size_t>((out_len + 63) / 64) * 8;
  if (allocate_validity) {
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], ctx->AllocateBitmap(out_len));
    ARROW_RETURN_NOT_OK(t_scratch.Get(kOutValidity, vbytes + 16, &d_out_valid));
  } else {
    out_arr->buffers[0] = nullptr;
  }
  ARROW_RETURN_NOT_OK(FromArx(arx_filter_exec(&dv, w, &dm, null_sel, ws, out_len, d_out, d_out_valid, st)));
  if (out_len > 0) {
    HIP_RETURN_NOT_OK(hipMemcpyAsync(out_arr->buffers[1]->mutable_data(), d_out,
                                     static_cast<size_t>(out_len) * w, hipMemcpyDeviceToHost, st));
    if (allocate_validity) {
      HIP_RETURN_NOT_OK(hipMemcpyAsync(out_arr->buffers[0]->mutable_data(), d_out_valid,
                                       static_cast<size_t>(arrow::bit_util::BytesForBits(out_len)),
                                       hipMemcpyDeviceToHost, st));
    }
  }
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  CountGpu(kFnFilter);
  return Status::OK();
}

int IndexTypeId(const arrow::DataType& t);

// ---------------------------------------------------------------- binary / utf8 (device-resident)
// TakeExec / FilterExec for base binary on arrays that live in HBM: the offsets/validity pass,
// one 8-byte read-back of the byte total (the reference grows a builder instead), then the bytes.
// Host-resident strings stay on the stock kernels (they would be PCIe-bound both ways).
StockKernel g_stock_filter_bin, g_stock_take_bin;

arrow::Result<std::unique_ptr<cp::KernelState>> BinaryFilterInit(cp::KernelContext* ctx,
                                                                 const cp::KernelInitArgs& args) {
  auto state = std::make_unique<ShimState<cp::FilterOptions>>();
  if (g_stock_filter_bin.init) {
    ARROW_ASSIGN_OR_RAISE(state->stock, g_stock_filter_bin.init(ctx, args));
  }
  if (args.options != nullptr) state->options = *static_cast<const cp::FilterOptions*>(args.options);
  return state;
}

arrow::Result<std::unique_ptr<cp::KernelState>> BinaryTakeInit(cp::KernelContext* ctx,
                                                               const cp::KernelInitArgs& args) {
  auto state = std::make_unique<ShimState<cp::TakeOptions>>();
  if (g_stock_take_bin.init) {
    ARROW_ASSIGN_OR_RAISE(state->stock, g_stock_take_bin.init(ctx, args));
  }
  if (args.options != nullptr) state->options = *static_cast<const cp::TakeOptions*>(args.options);
  return state;
}

bool OnRocm3(const ArraySpan& a) {
  if (OnRocm(a)) return true;
  return a.buffers[2].owner != nullptr && *a.buffers[2].owner != nullptr &&
         (*a.buffers[2].owner)->device_type() == arrow::DeviceAllocationType::kROCM;
}

Status DeviceBinarySpan(const ArraySpan& a, ArxBinarySpan* out) {
  const void* addr[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < 3; ++i) {
    const auto* owner = a.buffers[i].owner;
    if (owner == nullptr || *owner == nullptr) continue;
    if ((*owner)->device_type() != arrow::DeviceAllocationType::kROCM) {
      return Status::Invalid("arrow_amd: mixed host / device buffers in one array");
    }
    addr[i] = reinterpret_cast<const void*>((*owner)->address());
  }
  out->validity = addr[0];
  out->offsets = static_cast<const int32_t*>(addr[1]);
  out->data = addr[2];
  out->offset = a.offset;
  out->length = a.length;
  out->null_count = addr[0] == nullptr ? 0 : (a.null_count > 0 ? a.null_count : arrow::kUnknownNullCount);
  return Status::OK();
}

Status BinaryTakeOnDevice(const ArxBinarySpan& dv, const ArxSpan& di, int tid, hipStream_t st, ArrayData* out_arr) {
  const int64_t m = di.length;
  const bool allocate_validity = dv.null_count != 0 || di.null_count != 0;
  out_arr->length = m;
  out_arr->buffers.resize(3);
  out_arr->buffers[0] = nullptr;
  ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice((m + 1) * 4));
  void* d_valid = nullptr;
  void* d_counter = nullptr;
  if (allocate_validity) {
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(((m + 63) / 64) * 8));
    d_valid = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
    ARROW_RETURN_NOT_OK(t_scratch.Get(kCounter, 64, &d_counter));
    HIP_RETURN_NOT_OK(hipMemsetAsync(d_counter, 0, 8, st));
  }
  const size_t ws_bytes = arx_binary_take_workspace_bytes(m);
  void* ws = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kBinWs, ws_bytes, &ws));
  int32_t* d_off = reinterpret_cast<int32_t*>(out_arr->buffers[1]->mutable_address());
  int64_t total = 0;
  ARROW_RETURN_NOT_OK(FromArx(arx_binary_take_offsets(&dv, &di, tid, ws, ws_bytes, d_off, d_valid,
                                                      static_cast<int64_t*>(d_counter), &total, st)));
  ARROW_ASSIGN_OR_RAISE(out_arr->buffers[2], AllocDevice(total));
  ARROW_RETURN_NOT_OK(FromArx(arx_binary_take_data(&dv, m, ws, ws_bytes, d_off, total,
                                                   reinterpret_cast<void*>(out_arr->buffers[2]->mutable_address()),
                                                   st)));
  int64_t valid_count = m;
  if (allocate_validity) {
    HIP_RETURN_NOT_OK(hipMemcpyAsync(&valid_count, d_counter, 8, hipMemcpyDeviceToHost, st));
  }
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  out_arr->null_count = m - valid_count;
  return Status::OK();
}

bool IsInt32Binary(const arrow::DataType& t) { return t.id() == Type::STRING || t.id() == Type::BINARY; }

Status BinaryTakeExec(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  auto* state = static_cast<ShimState<cp::TakeOptions>*>(ctx->state());
  const ArraySpan& values = batch[0].array;
  const ArraySpan& indices = batch[1].array;
  if (!OnRocm3(values) && !OnRocm(indices)) {
    return RunStock(kFnTake, g_stock_take_bin, state->stock.get(), ctx, batch, out);
  }
  const int tid = IndexTypeId(*indices.type);
  if (tid < 0 || !IsInt32Binary(*values.type)) {
    return Status::NotImplemented("arrow_amd: take of ", values.type->ToString(), " by ",
                                  indices.type->ToString(), " on device-resident arrays");
  }
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxBinarySpan dv{};
  ArxSpan di{};
  ARROW_RETURN_NOT_OK(DeviceBinarySpan(values, &dv));
  ARROW_RETURN_NOT_OK(DeviceSpan(indices, &di));
  if (state->options.boundscheck) {
    void* ws = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, arx_take_workspace_bytes(), &ws));
    ARROW_RETURN_NOT_OK(FromArx(arx_check_index_bounds(&di, tid, static_cast<uint64_t>(values.length), ws,
                                                       arx_take_workspace_bytes(), st)));
  }
  ARROW_RETURN_NOT_OK(BinaryTakeOnDevice(dv, di, tid, st, out->array_data().get()));
  CountGpu(kFnTake);
  return Status::OK();
}

// BinaryFilterImpl == take(GetTakeIndices(filter)): the indices live in device scratch only
Status BinaryFilterExec(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  auto* state = static_cast<ShimState<cp::FilterOptions>*>(ctx->state());
  const ArraySpan& values = batch[0].array;
  const ArraySpan& filter = batch[1].array;
  if (!OnRocm3(values) && !OnRocm(filter)) {
    return RunStock(kFnFilter, g_stock_filter_bin, state->stock.get(), ctx, batch, out);
  }
  if (filter.type->id() != Type::BOOL || !IsInt32Binary(*values.type)) {
    return Status::NotImplemented("arrow_amd: filter of ", values.type->ToString(), " on device-resident arrays");
  }
  if (filter.length > 0xFFFFFFFFll) {
    return Status::NotImplemented("Filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");
  }
  const int null_sel = state->options.null_selection_behavior == cp::FilterOptions::EMIT_NULL
                           ? ARX_FILTER_EMIT_NULL : ARX_FILTER_DROP;
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxBinarySpan dv{};
  ArxSpan dm{};
  ARROW_RETURN_NOT_OK(DeviceBinarySpan(values, &dv));
  ARROW_RETURN_NOT_OK(DeviceSpan(filter, &dm));
  const size_t ws_bytes = arx_filter_workspace_bytes(filter.length);
  void* ws = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes, &ws));
  int64_t out_len = 0;
  ARROW_RETURN_NOT_OK(FromArx(arx_filter_count(&dm, null_sel, ws, ws_bytes, &out_len, st)));
  const bool emit = null_sel == ARX_FILTER_EMIT_NULL && dm.null_count != 0;
  void* d_idx = nullptr;
  void* d_idx_valid = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2, static_cast<size_t>(out_len) * 4 + 64, &d_idx));
  if (emit) ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2Validity, static_cast<size_t>((out_len + 63) / 64) * 8 + 64, &d_idx_valid));
  ARROW_RETURN_NOT_OK(FromArx(arx_mask_to_indices(&dm, null_sel, ws, out_len, 4, d_idx, d_idx_valid, st)));
  ArxSpan di{d_idx_valid, d_idx, 0, out_len, emit ? arrow::kUnknownNullCount : 0};
  ARROW_RETURN_NOT_OK(BinaryTakeOnDevice(dv, di, ARX_UINT32, st, out->array_data().get()));
  CountGpu(kFnFilter);
  return Status::OK();
}

// ---------------------------------------------------------------- take
StockKernel g_stock_take;

arrow::Result<std::unique_ptr<cp::KernelState>> TakeInit(cp::KernelContext* ctx,
                                                         const cp::KernelInitArgs& args) {
  auto state = std::make_unique<ShimState<cp::TakeOptions>>();
  if (g_stock_take.init) {
    ARROW_ASSIGN_OR_RAISE(state->stock, g_stock_take.init(ctx, args));
  }
  if (args.options != nullptr) state->options = *static_cast<const cp::TakeOptions*>(args.options);
  return state;
}

int IndexTypeId(const arrow::DataType& t) {
  switch (t.id()) {
    case Type::UINT8: return ARX_UINT8;
    case Type::INT8: return ARX_INT8;
    case Type::UINT16: return ARX_UINT16;
    case Type::INT16: return ARX_INT16;
    case Type::UINT32: return ARX_UINT32;
    case Type::INT32: return ARX_INT32;
    case Type::UINT64: return ARX_UINT64;
    case Type::INT64: return ARX_INT64;
    default: return -1;
  }
}

// FixedWidthTakeExec (vector_selection_take_internal.cc:405-468) with the gather on the GPU.
Status TakeExec(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  auto* state = static_cast<ShimState<cp::TakeOptions>*>(ctx->state());
  const ArraySpan& values = batch[0].array;
  const ArraySpan& indices = batch[1].array;
  const int w = FixedByteWidth(*values.type);
  const int tid = IndexTypeId(*indices.type);
  static const int kIdxWidth[8] = {1, 1, 2, 2, 4, 4, 8, 8};
  if (OnRocm(values) || OnRocm(indices)) {
    if (w == 0 || tid < 0) {
      return Status::NotImplemented("arrow_amd: take of ", values.type->ToString(), " by ",
                                    indices.type->ToString(), " on device-resident arrays");
    }
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ArxSpan dv{}, di{};
    ARROW_RETURN_NOT_OK(DeviceSpan(values, &dv));
    ARROW_RETURN_NOT_OK(DeviceSpan(indices, &di));
    if (state->options.boundscheck) {
      void* ws = nullptr;
      ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, arx_take_workspace_bytes(), &ws));
      ARROW_RETURN_NOT_OK(FromArx(arx_check_index_bounds(&di, tid, static_cast<uint64_t>(values.length), ws,
                                                         arx_take_workspace_bytes(), st)));
    }
    const int64_t m = indices.length;
    const bool allocate_validity = dv.null_count != 0 || di.null_count != 0;
    ArrayData* out_arr = out->array_data().get();
    out_arr->length = m;
    out_arr->buffers.resize(2);
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(m * w));
    out_arr->buffers[0] = nullptr;
    void* d_valid = nullptr;
    void* d_counter = nullptr;
    if (allocate_validity) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(((m + 63) / 64) * 8));
      d_valid = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
      ARROW_RETURN_NOT_OK(t_scratch.Get(kCounter, 64, &d_counter));
      HIP_RETURN_NOT_OK(hipMemsetAsync(d_counter, 0, 8, st));
    }
    ARROW_RETURN_NOT_OK(FromArx(arx_take(&dv, w, &di, tid,
                                         reinterpret_cast<void*>(out_arr->buffers[1]->mutable_address()), d_valid,
                                         static_cast<int64_t*>(d_counter), st)));
    int64_t valid_count = m;
    if (allocate_validity) {
      HIP_RETURN_NOT_OK(hipMemcpyAsync(&valid_count, d_counter, 8, hipMemcpyDeviceToHost, st));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    out_arr->null_count = m - valid_count;
    CountGpu(kFnTake);
    return Status::OK();
  }
  if (w == 0 || tid < 0 || indices.length < g_min_rows.load() || !IsHost(values) || !IsHost(indices)) {
    return RunStock(kFnTake, g_stock_take, state->stock.get(), ctx, batch, out);
  }
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxSpan dv{}, di{};
  ARROW_RETURN_NOT_OK(Upload(values, w, kValues, kValidity, st, &dv));
  ARROW_RETURN_NOT_OK(Upload(indices, kIdxWidth[tid], kArg2, kArg2Validity, st, &di));
  if (state->options.boundscheck) {
    void* ws = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, arx_take_workspace_bytes(), &ws));
    ARROW_RETURN_NOT_OK(FromArx(arx_check_index_bounds(&di, tid, static_cast<uint64_t>(values.length), ws,
                                                       arx_take_workspace_bytes(), st)));
  }
  const int64_t m = indices.length;
  const bool allocate_validity = values.MayHaveNulls() || indices.MayHaveNulls();
  ArrayData* out_arr = out->array_data().get();
  out_arr->length = m;
  out_arr->buffers.resize(2);
  ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], ctx->Allocate(m * w));
  void* d_out = nullptr;
  void* d_out_valid = nullptr;
  void* d_counter = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, static_cast<size_t>(m) * w + 16, &d_out));
  if (allocate_validity) {
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], ctx->AllocateBitmap(m));
    ARROW_RETURN_NOT_OK(t_scratch.Get(kOutValidity, static_cast<size_t>((m + 63) / 64) * 8 + 16, &d_out_valid));
    ARROW_RETURN_NOT_OK(t_scratch.Get(kCounter, 64, &d_counter));
    HIP_RETURN_NOT_OK(hipMemsetAsync(d_counter, 0, 8, st));
  } else {
    out_arr->buffers[0] = nullptr;
  }
  ARROW_RETURN_NOT_OK(FromArx(arx_take(&dv, w, &di, tid, d_out, d_out_valid,
                                       static_cast<int64_t*>(d_counter), st)));
  HIP_RETURN_NOT_OK(hipMemcpyAsync(out_arr->buffers[1]->mutable_data(), d_out, static_cast<size_t>(m) * w,
                                   hipMemcpyDeviceToHost, st));
  int64_t valid_count = m;
  if (allocate_validity) {
    HIP_RETURN_NOT_OK(hipMemcpyAsync(out_arr->buffers[0]->mutable_data(), d_out_valid,
                                     static_cast<size_t>(arrow::bit_util::BytesForBits(m)),
                                     hipMemcpyDeviceToHost, st));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(&valid_count, d_counter, 8, hipMemcpyDeviceToHost, st));
  }
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  out_arr->null_count = m - valid_count;
  CountGpu(kFnTake);
  return Status::OK();
}

// ---------------------------------------------------------------- greater(double, double)
StockKernel g_stock_greater;

// ComparePrimitiveArrayArray<DoubleType, Greater> (scalar_compare.cc:165-190); validity is
// handled by the ScalarExecutor (NullHandling::INTERSECTION), the output bitmap is preallocated.
Status GreaterExec(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  if (!batch[0].is_array() || !batch[1].is_array() || !out->is_array_span() ||
      out->array_span()->offset != 0 || batch.length < g_min_rows_streaming.load() ||
      !IsHost(batch[0].array) || !IsHost(batch[1].array)) {
    CountStock(kFnGreater);
    return g_stock_greater.exec(ctx, batch, out);
  }
  const ArraySpan& l = batch[0].array;
  const ArraySpan& r = batch[1].array;
  const int64_t n = batch.length;
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  void *dl = nullptr, *dr = nullptr, *dout = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kValues, static_cast<size_t>(n) * 8, &dl));
  ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2, static_cast<size_t>(n) * 8, &dr));
  ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, static_cast<size_t>((n + 63) / 64) * 8, &dout));
  HIP_RETURN_NOT_OK(hipMemcpyAsync(dl, l.GetValues<double>(1), static_cast<size_t>(n) * 8, hipMemcpyHostToDevice, st));
  HIP_RETURN_NOT_OK(hipMemcpyAsync(dr, r.GetValues<double>(1), static_cast<size_t>(n) * 8, hipMemcpyHostToDevice, st));
  ARROW_RETURN_NOT_OK(FromArx(arx_greater_f64(static_cast<const double*>(dl), static_cast<const double*>(dr), n,
                                              static_cast<uint64_t*>(dout), st)));
  ArraySpan* o = out->array_span_mutable();
  HIP_RETURN_NOT_OK(hipMemcpyAsync(o->buffers[1].data, dout, static_cast<size_t>(arrow::bit_util::BytesForBits(n)),
                                   hipMemcpyDeviceToHost, st));
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  CountGpu(kFnGreater);
  return Status::OK();
}

// The kernel actually registered for greater(double, double): NullHandling::COMPUTED_NO_PREALLOCATE +
// MemAllocation::NO_PREALLOCATE, because the ScalarExecutor's own preallocation and null
// propagation (exec.cc:846-861,1222-1281) run on the CPU and cannot touch device buffers.
//  * device-resident inputs: compare, validity intersection and null count all on the MI355X,
//    output bitmap + validity stay in HBM (so compare -> filter chains never leave the device);
//  * host inputs: allocate what the executor would have preallocated, propagate nulls with Arrow's
//    own bitmap utilities, then run the preallocated-style exec above (HIP staging path for large
//    arrays, Arrow's stock kernel for scalars / small inputs).
Status GreaterExecNP(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  const int64_t n = batch.length;
  ArrayData* out_arr = out->array_data().get();
  out_arr->buffers.resize(2);
  const bool dev0 = batch[0].is_array() && OnRocm(batch[0].array);
  const bool dev1 = batch[1].is_array() && OnRocm(batch[1].array);
  if (dev0 || dev1) {
    for (int i = 0; i < 2; ++i) {
      if (batch[i].is_array() ? !OnRocm(batch[i].array) : !batch[i].scalar->is_valid) {
        return Status::NotImplemented("arrow_amd: greater on device-resident arrays needs device arrays or "
                                      "valid scalars on both sides");
      }
    }
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(((n + 63) / 64) * 8));
    uint64_t* dout = reinterpret_cast<uint64_t*>(out_arr->buffers[1]->mutable_address());
    ArxSpan sp[2] = {};
    const double* ptr[2] = {nullptr, nullptr};
    double sc[2] = {0.0, 0.0};
    for (int i = 0; i < 2; ++i) {
      if (batch[i].is_array()) {
        ARROW_RETURN_NOT_OK(DeviceSpan(batch[i].array, &sp[i]));
        ptr[i] = static_cast<const double*>(sp[i].data) + sp[i].offset;
      } else {
        sc[i] = static_cast<const arrow::DoubleScalar&>(*batch[i].scalar).value;
      }
    }
    int rc;
    if (ptr[0] && ptr[1]) rc = arx_greater_f64(ptr[0], ptr[1], n, dout, st);
    else if (ptr[0]) rc = arx_greater_f64_array_scalar(ptr[0], sc[1], n, dout, st);
    else rc = arx_greater_f64_scalar_array(sc[0], ptr[1], n, dout, st);
    ARROW_RETURN_NOT_OK(FromArx(rc));
    // validity = intersection of the inputs' validity bitmaps, re-based to offset 0
    const ArxSpan* with_nulls[2];
    int nv = 0;
    for (int i = 0; i < 2; ++i) {
      if (ptr[i] && sp[i].validity != nullptr) with_nulls[nv++] = &sp[i];
    }
    out_arr->buffers[0] = nullptr;
    out_arr->null_count = 0;
    if (nv > 0 && n > 0) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(((n + 63) / 64) * 8));
      void* dv = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
      if (nv == 1) {
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_copy(with_nulls[0]->validity, with_nulls[0]->offset, n, dv, st)));
      } else {
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_and(with_nulls[0]->validity, with_nulls[0]->offset,
                                                   with_nulls[1]->validity, with_nulls[1]->offset, n, dv, st)));
      }
      ARROW_ASSIGN_OR_RAISE(out_arr->null_count, DeviceNullCount(*out_arr->buffers[0], n, st));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnGreater);
    return Status::OK();
  }

  // ---- host inputs: what ScalarExecutor::PrepareOutput + PropagateNulls would have done
  arrow::MemoryPool* pool = ctx->memory_pool();
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> data, ctx->AllocateBitmap(n));
  std::shared_ptr<Buffer> validity;
  int64_t null_count = 0;
  bool null_scalar = false;
  const ArraySpan* with_nulls[2];
  int nv = 0;
  for (int i = 0; i < 2; ++i) {
    if (batch[i].is_scalar()) {
      null_scalar = null_scalar || !batch[i].scalar->is_valid;
    } else if (batch[i].array.MayHaveNulls()) {
      with_nulls[nv++] = &batch[i].array;
    }
  }
  if (null_scalar) {
    ARROW_ASSIGN_OR_RAISE(validity, ctx->AllocateBitmap(n));  // zero-initialised: every slot null
    null_count = n;
  } else if (nv == 1) {
    ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::CopyBitmap(pool, with_nulls[0]->buffers[0].data,
                                                                with_nulls[0]->offset, n));
    null_count = with_nulls[0]->null_count;  // may be kUnknownNullCount
  } else if (nv == 2) {
    ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::BitmapAnd(pool, with_nulls[0]->buffers[0].data,
                                                               with_nulls[0]->offset, with_nulls[1]->buffers[0].data,
                                                               with_nulls[1]->offset, n, 0));
    null_count = arrow::kUnknownNullCount;
  }
  cp::ExecResult tmp;
  ArraySpan span;
  span.type = out_arr->type.get();
  span.length = n;
  span.offset = 0;
  span.null_count = null_count;
  if (validity) {
    span.buffers[0].data = validity->mutable_data();
    span.buffers[0].size = validity->size();
  }
  span.buffers[1].data = data->mutable_data();
  span.buffers[1].size = data->size();
  tmp.value = std::move(span);
  ARROW_RETURN_NOT_OK(GreaterExec(ctx, batch, &tmp));
  out_arr->buffers[0] = std::move(validity);
  out_arr->buffers[1] = std::move(data);
  out_arr->null_count = null_count;
  return Status::OK();
}

// ---------------------------------------------------------------- greater(int64), add(int64|double)
// The same NO_PREALLOCATE twin as GreaterExecNP, generic over the operation: device-resident
// operands (arrays, or one valid scalar) run on the MI355X and the result stays in HBM; host
// operands get exactly the buffers the ScalarExecutor would have preallocated and then go to
// Arrow's stock kernel (these element-wise ops are PCIe-bound for host data).
StockKernel g_stock_greater_i64, g_stock_add_i64, g_stock_add_f64;

struct OpGreaterI64 {
  static constexpr bool kChecked = false;
  using T = int64_t;
  using ScalarT = arrow::Int64Scalar;
  static constexpr bool kBitmapOut = true;
  static constexpr Fn kFn = kFnGreater;
  static StockKernel& stock() { return g_stock_greater_i64; }
  static int aa(const T* l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_greater_i64(l, r, n, static_cast<uint64_t*>(o), st); }
  static int as(const T* l, T r, int64_t n, void* o, hipStream_t st) { return arx_greater_i64_array_scalar(l, r, n, static_cast<uint64_t*>(o), st); }
  static int sa(T l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_greater_i64_scalar_array(l, r, n, static_cast<uint64_t*>(o), st); }
};
struct OpAddI64 {
  static constexpr bool kChecked = false;
  using T = int64_t;
  using ScalarT = arrow::Int64Scalar;
  static constexpr bool kBitmapOut = false;
  static constexpr Fn kFn = kFnAdd;
  static StockKernel& stock() { return g_stock_add_i64; }
  static int aa(const T* l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_add_i64(l, r, n, static_cast<T*>(o), st); }
  static int as(const T* l, T r, int64_t n, void* o, hipStream_t st) { return arx_add_i64_array_scalar(l, r, n, static_cast<T*>(o), st); }
  static int sa(T l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_add_i64_array_scalar(r, l, n, static_cast<T*>(o), st); }
};
struct OpAddF64 {
  static constexpr bool kChecked = false;
  using T = double;
  using ScalarT = arrow::DoubleScalar;
  static constexpr bool kBitmapOut = false;
  static constexpr Fn kFn = kFnAdd;
  static StockKernel& stock() { return g_stock_add_f64; }
  static int aa(const T* l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_add_f64(l, r, n, static_cast<T*>(o), st); }
  static int as(const T* l, T r, int64_t n, void* o, hipStream_t st) { return arx_add_f64_array_scalar(l, r, n, static_cast<T*>(o), st); }
  static int sa(T l, const T* r, int64_t n, void* o, hipStream_t st) { return arx_add_f64_array_scalar(r, l, n, static_cast<T*>(o), st); }
};

// equal / not_equal / greater_equal / less / less_equal for int64 and double (greater has its own
// entries above): one stock-kernel slot per (function, type)
StockKernel g_stock_compare[12];

template <typename CT, typename ScalarType, int CMP, int SLOT>
struct OpCompare {
  using T = CT;
  using ScalarT = ScalarType;
  static constexpr bool kBitmapOut = true;
  static constexpr bool kChecked = false;
  static constexpr Fn kFn = kFnCompare;
  static StockKernel& stock() { return g_stock_compare[SLOT]; }
  static int run(const T* l, T ls, const T* r, T rs, int64_t n, void* o, hipStream_t st) {
    if constexpr (std::is_same<T, double>::value) {
      return arx_compare_f64(CMP, l, ls, r, rs, n, static_cast<uint64_t*>(o), st);
    } else {
      return arx_compare_i64(CMP, l, ls, r, rs, n, static_cast<uint64_t*>(o), st);
    }
  }
  static int aa(const T* l, const T* r, int64_t n, void* o, hipStream_t st) { return run(l, T(0), r, T(0), n, o, st); }
  static int as(const T* l, T r, int64_t n, void* o, hipStream_t st) { return run(l, T(0), nullptr, r, n, o, st); }
  static int sa(T l, const T* r, int64_t n, void* o, hipStream_t st) { return run(nullptr, l, r, T(0), n, o, st); }
};

// subtract / multiply and add_checked / subtract_checked / multiply_checked (int64, double): what
// `+`, `-`, `*` on pyarrow / Acero expressions mean.  The checked int64 forms read an overflow flag
// back after the kernel and fail with the reference's Status::Invalid("overflow").
StockKernel g_stock_arith[10];

template <typename CT, typename ScalarType, int OP, bool CHECKED, int SLOT>
struct OpArith {
  using T = CT;
  using ScalarT = ScalarType;
  static constexpr bool kBitmapOut = false;
  static constexpr bool kChecked = CHECKED && std::is_same<CT, int64_t>::value;
  static constexpr Fn kFn = kFnAdd;
  static StockKernel& stock() { return g_stock_arith[SLOT]; }
  static int run(const T* l, T ls, const T* r, T rs, int64_t n, void* o, hipStream_t st) {
    if constexpr (std::is_same<T, double>::value) {
      return arx_arith_f64(OP, l, ls, r, rs, n, static_cast<double*>(o), st);
    } else {
      return arx_arith_i64(OP, l, ls, r, rs, n, static_cast<int64_t*>(o), st);
    }
  }
  static int aa(const T* l, const T* r, int64_t n, void* o, hipStream_t st) { return run(l, T(0), r, T(0), n, o, st); }
  static int as(const T* l, T r, int64_t n, void* o, hipStream_t st) { return run(l, T(0), nullptr, r, n, o, st); }
  static int sa(T l, const T* r, int64_t n, void* o, hipStream_t st) { return run(nullptr, l, r, T(0), n, o, st); }
  static int checked(const T* l, T ls, const ArxSpan& lsp, const T* r, T rs, const ArxSpan& rsp, int64_t n, void* o,
                     unsigned int* flag, hipStream_t st) {
    if constexpr (std::is_same<T, int64_t>::value) {
      return arx_arith_checked_i64(OP, l, ls, l ? lsp.validity : nullptr, lsp.offset, r, rs,
                                   r ? rsp.validity : nullptr, rsp.offset, n, static_cast<int64_t*>(o), flag, st);
    } else {
      return ARX_NOT_IMPLEMENTED;
    }
  }
};

template <class Op>
Status ScalarBinaryNP(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  using T = typename Op::T;
  const int64_t n = batch.length;
  ArrayData* out_arr = out->array_data().get();
  out_arr->buffers.resize(2);
  const bool dev0 = batch[0].is_array() && OnRocm(batch[0].array);
  const bool dev1 = batch[1].is_array() && OnRocm(batch[1].array);
  const int64_t data_bytes = Op::kBitmapOut ? ((n + 63) / 64) * 8 : n * static_cast<int64_t>(sizeof(T));
  if (dev0 || dev1) {
    for (int i = 0; i < 2; ++i) {
      if (batch[i].is_array() ? !OnRocm(batch[i].array) : !batch[i].scalar->is_valid) {
        return Status::NotImplemented("arrow_amd: ", kFnNames[Op::kFn], " on device-resident arrays needs device "
                                      "arrays or valid scalars on both sides");
      }
    }
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(data_bytes));
    void* dout = reinterpret_cast<void*>(out_arr->buffers[1]->mutable_address());
    ArxSpan sp[2] = {};
    const T* ptr[2] = {nullptr, nullptr};
    T sc[2] = {T(0), T(0)};
    for (int i = 0; i < 2; ++i) {
      if (batch[i].is_array()) {
        ARROW_RETURN_NOT_OK(DeviceSpan(batch[i].array, &sp[i]));
        ptr[i] = static_cast<const T*>(sp[i].data) + sp[i].offset;
      } else {
        sc[i] = static_cast<const typename Op::ScalarT&>(*batch[i].scalar).value;
      }
    }
    int rc;
    unsigned int* d_flag = nullptr;
    if constexpr (Op::kChecked) {
      void* f = nullptr;
      ARROW_RETURN_NOT_OK(t_scratch.Get(kFlag, 64, &f));
      HIP_RETURN_NOT_OK(hipMemsetAsync(f, 0, 4, st));
      d_flag = static_cast<unsigned int*>(f);
      rc = Op::checked(ptr[0], sc[0], sp[0], ptr[1], sc[1], sp[1], n, dout, d_flag, st);
    } else {
      if (ptr[0] && ptr[1]) rc = Op::aa(ptr[0], ptr[1], n, dout, st);
      else if (ptr[0]) rc = Op::as(ptr[0], sc[1], n, dout, st);
      else rc = Op::sa(sc[0], ptr[1], n, dout, st);
    }
    ARROW_RETURN_NOT_OK(FromArx(rc));
    const ArxSpan* with_nulls[2];
    int nv = 0;
    for (int i = 0; i < 2; ++i) {
      if (ptr[i] && sp[i].validity != nullptr) with_nulls[nv++] = &sp[i];
    }
    out_arr->buffers[0] = nullptr;
    out_arr->null_count = 0;
    if (nv > 0 && n > 0) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(((n + 63) / 64) * 8));
      void* dv = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
      if (nv == 1) {
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_copy(with_nulls[0]->validity, with_nulls[0]->offset, n, dv, st)));
      } else {
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_and(with_nulls[0]->validity, with_nulls[0]->offset,
                                                   with_nulls[1]->validity, with_nulls[1]->offset, n, dv, st)));
      }
      ARROW_ASSIGN_OR_RAISE(out_arr->null_count, DeviceNullCount(*out_arr->buffers[0], n, st));
    }
    unsigned int flag = 0;
    if (d_flag != nullptr) HIP_RETURN_NOT_OK(hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st));
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    if (flag != 0) return Status::Invalid("overflow");  // AddChecked::Call, base_arithmetic_internal.h:77
    CountGpu(Op::kFn);
    return Status::OK();
  }

  // ---- host operands: ScalarExecutor::PrepareOutput + PropagateNulls, then Arrow's stock kernel
  arrow::MemoryPool* pool = ctx->memory_pool();
  std::shared_ptr<Buffer> data;
  if (Op::kBitmapOut) {
    ARROW_ASSIGN_OR_RAISE(data, ctx->AllocateBitmap(n));
  } else {
    ARROW_ASSIGN_OR_RAISE(data, ctx->Allocate(data_bytes));
  }
  std::shared_ptr<Buffer> validity;
  int64_t null_count = 0;
  bool null_scalar = false;
  const ArraySpan* with_nulls[2];
  int nv = 0;
  for (int i = 0; i < 2; ++i) {
    if (batch[i].is_scalar()) {
      null_scalar = null_scalar || !batch[i].scalar->is_valid;
    } else if (batch[i].array.MayHaveNulls()) {
      with_nulls[nv++] = &batch[i].array;
    }
  }
  if (null_scalar) {
    ARROW_ASSIGN_OR_RAISE(validity, ctx->AllocateBitmap(n));
    null_count = n;
  } else if (nv == 1) {
    ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::CopyBitmap(pool, with_nulls[0]->buffers[0].data,
                                                                with_nulls[0]->offset, n));
    null_count = with_nulls[0]->null_count;
  } else if (nv == 2) {
    ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::BitmapAnd(pool, with_nulls[0]->buffers[0].data,
                                                               with_nulls[0]->offset, with_nulls[1]->buffers[0].data,
                                                               with_nulls[1]->offset, n, 0));
    null_count = arrow::kUnknownNullCount;
  }
  cp::ExecResult tmp;
  ArraySpan span;
  span.type = out_arr->type.get();
  span.length = n;
  span.offset = 0;
  span.null_count = null_count;
  if (validity) {
    span.buffers[0].data = validity->mutable_data();
    span.buffers[0].size = validity->size();
  }
  span.buffers[1].data = data->mutable_data();
  span.buffers[1].size = data->size();
  tmp.value = std::move(span);
  CountStock(Op::kFn);
  ARROW_RETURN_NOT_OK(Op::stock().exec(ctx, batch, &tmp));
  out_arr->buffers[0] = std::move(validity);
  out_arr->buffers[1] = std::move(data);
  out_arr->null_count = null_count;
  return Status::OK();
}

template <class Op>
Status RegisterScalarBinaryNP(cp::FunctionRegistry* reg, const char* name, const std::shared_ptr<arrow::DataType>& t) {
  ARROW_ASSIGN_OR_RAISE(auto fn, reg->GetFunction(name));
  auto* sfn = static_cast<cp::ScalarFunction*>(fn.get());
  ARROW_ASSIGN_OR_RAISE(const cp::Kernel* k0, sfn->DispatchExact({t, t}));
  cp::ScalarKernel copy = *static_cast<const cp::ScalarKernel*>(k0);
  Op::stock().exec = copy.exec;
  Op::stock().init = copy.init;
  copy.signature = cp::KernelSignature::Make({cp::InputType(t), cp::InputType(t)}, copy.signature->out_type());
  copy.exec = ScalarBinaryNP<Op>;
  copy.null_handling = cp::NullHandling::COMPUTED_NO_PREALLOCATE;
  copy.mem_allocation = cp::MemAllocation::NO_PREALLOCATE;
  return sfn->AddKernel(std::move(copy));
}

// ---------------------------------------------------------------- and_kleene / or_kleene / invert
// What Acero filter expressions like (a > 1) & (b > 2) evaluate to.  Same NO_PREALLOCATE twin:
// device-resident boolean arrays run arx_boolean_kleene / arx_boolean_invert and stay in HBM; host
// operands get the buffers the executor would have preallocated (data + validity bitmaps) and go to
// Arrow's stock kernel.
StockKernel g_stock_and_kleene, g_stock_or_kleene, g_stock_invert;

template <int OP>
Status KleeneExecNP(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  StockKernel& stock = OP == ARX_AND_KLEENE ? g_stock_and_kleene : g_stock_or_kleene;
  const int64_t n = batch.length;
  ArrayData* out_arr = out->array_data().get();
  out_arr->buffers.resize(2);
  const bool dev0 = batch[0].is_array() && OnRocm(batch[0].array);
  const bool dev1 = batch[1].is_array() && OnRocm(batch[1].array);
  if (dev0 || dev1) {
    if (!(dev0 && dev1)) {
      return Status::NotImplemented("arrow_amd: and_kleene / or_kleene on device-resident arrays needs device "
                                    "arrays on both sides");
    }
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ArxSpan l{}, r{};
    ARROW_RETURN_NOT_OK(DeviceSpan(batch[0].array, &l));
    ARROW_RETURN_NOT_OK(DeviceSpan(batch[1].array, &r));
    const int64_t bytes = ((n + 63) / 64) * 8;
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(bytes));
    out_arr->buffers[0] = nullptr;
    void* dvalid = nullptr;
    const bool nulls = l.validity != nullptr || r.validity != nullptr;
    if (nulls) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(bytes));
      dvalid = reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address());
    }
    ARROW_RETURN_NOT_OK(FromArx(arx_boolean_kleene(OP, &l, &r, reinterpret_cast<void*>(out_arr->buffers[1]->mutable_address()),
                                                   dvalid, st)));
    out_arr->null_count = 0;
    if (nulls && n > 0) {
      ARROW_ASSIGN_OR_RAISE(out_arr->null_count, DeviceNullCount(*out_arr->buffers[0], n, st));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnBoolean);
    return Status::OK();
  }
  // host operands: NullHandling::COMPUTED_PREALLOCATE + MemAllocation::PREALLOCATE of the stock kernel
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> data, ctx->AllocateBitmap(n));
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> validity, ctx->AllocateBitmap(n));
  cp::ExecResult tmp;
  ArraySpan span;
  span.type = out_arr->type.get();
  span.length = n;
  span.offset = 0;
  span.null_count = arrow::kUnknownNullCount;
  span.buffers[0].data = validity->mutable_data();
  span.buffers[0].size = validity->size();
  span.buffers[1].data = data->mutable_data();
  span.buffers[1].size = data->size();
  tmp.value = std::move(span);
  CountStock(kFnBoolean);
  ARROW_RETURN_NOT_OK(stock.exec(ctx, batch, &tmp));
  out_arr->null_count = tmp.array_span()->null_count;
  out_arr->buffers[0] = std::move(validity);
  out_arr->buffers[1] = std::move(data);
  return Status::OK();
}

Status InvertExecNP(cp::KernelContext* ctx, const cp::ExecSpan& batch, cp::ExecResult* out) {
  const int64_t n = batch.length;
  ArrayData* out_arr = out->array_data().get();
  out_arr->buffers.resize(2);
  if (batch[0].is_array() && OnRocm(batch[0].array)) {
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ArxSpan a{};
    ARROW_RETURN_NOT_OK(DeviceSpan(batch[0].array, &a));
    const int64_t bytes = ((n + 63) / 64) * 8;
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(bytes));
    ARROW_RETURN_NOT_OK(FromArx(arx_boolean_invert(a.data, a.offset, n,
                                                   reinterpret_cast<void*>(out_arr->buffers[1]->mutable_address()), st)));
    out_arr->buffers[0] = nullptr;
    out_arr->null_count = 0;
    if (a.validity != nullptr && n > 0) {
      ARROW_ASSIGN_OR_RAISE(out_arr->buffers[0], AllocDevice(bytes));
      ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_copy(a.validity, a.offset, n,
                                                  reinterpret_cast<void*>(out_arr->buffers[0]->mutable_address()), st)));
      ARROW_ASSIGN_OR_RAISE(out_arr->null_count, DeviceNullCount(*out_arr->buffers[0], n, st));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnBoolean);
    return Status::OK();
  }
  // host operand: NullHandling::INTERSECTION + PREALLOCATE
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> data, ctx->AllocateBitmap(n));
  std::shared_ptr<Buffer> validity;
  int64_t null_count = 0;
  if (batch[0].is_scalar()) {
    if (!batch[0].scalar->is_valid) {
      ARROW_ASSIGN_OR_RAISE(validity, ctx->AllocateBitmap(n));
      null_count = n;
    }
  } else if (batch[0].array.MayHaveNulls()) {
    const ArraySpan& a = batch[0].array;
    ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::CopyBitmap(ctx->memory_pool(), a.buffers[0].data, a.offset, n));
    null_count = a.null_count;
  }
  cp::ExecResult tmp;
  ArraySpan span;
  span.type = out_arr->type.get();
  span.length = n;
  span.offset = 0;
  span.null_count = null_count;
  if (validity) {
    span.buffers[0].data = validity->mutable_data();
    span.buffers[0].size = validity->size();
  }
  span.buffers[1].data = data->mutable_data();
  span.buffers[1].size = data->size();
  tmp.value = std::move(span);
  CountStock(kFnBoolean);
  ARROW_RETURN_NOT_OK(g_stock_invert.exec(ctx, batch, &tmp));
  out_arr->buffers[0] = std::move(validity);
  out_arr->buffers[1] = std::move(data);
  out_arr->null_count = null_count;
  return Status::OK();
}

Status RegisterBooleanNP(cp::FunctionRegistry* reg, const char* name, int arity, cp::ArrayKernelExec exec,
                         StockKernel* stock) {
  ARROW_ASSIGN_OR_RAISE(auto fn, reg->GetFunction(name));
  auto* sfn = static_cast<cp::ScalarFunction*>(fn.get());
  std::vector<arrow::TypeHolder> types(arity, arrow::boolean());
  ARROW_ASSIGN_OR_RAISE(const cp::Kernel* k0, sfn->DispatchExact(types));
  cp::ScalarKernel copy = *static_cast<const cp::ScalarKernel*>(k0);
  stock->exec = copy.exec;
  stock->init = copy.init;
  copy.exec = exec;
  copy.null_handling = cp::NullHandling::COMPUTED_NO_PREALLOCATE;
  copy.mem_allocation = cp::MemAllocation::NO_PREALLOCATE;
  return sfn->AddKernel(std::move(copy));
}

// ---------------------------------------------------------------- array_sort_indices(uint64|int64)
// One slot per registered value type: 0-5 = the ARX_KEY_* types themselves, 6-11 = temporal types
// sorted by their physical integer (date32, date64, timestamp, duration, time32, time64).
constexpr int kSortSlots = 12;
constexpr int kSortSlotKey[kSortSlots] = {ARX_KEY_UINT64, ARX_KEY_INT64, ARX_KEY_UINT32, ARX_KEY_INT32,
                                          ARX_KEY_FLOAT64, ARX_KEY_FLOAT32, ARX_KEY_INT32, ARX_KEY_INT64,
                                          ARX_KEY_INT64, ARX_KEY_INT64, ARX_KEY_INT32, ARX_KEY_INT64};
StockKernel g_stock_sort[kSortSlots];

arrow::Result<std::unique_ptr<cp::KernelState>> SortInitImpl(const StockKernel& stock,
                                                             cp::KernelContext* ctx,
                                                             const cp::KernelInitArgs& args) {
  auto state = std::make_unique<ShimState<cp::ArraySortOptions>>();
  if (stock.init) {
    ARROW_ASSIGN_OR_RAISE(state->stock, stock.init(ctx, args));
  }
  if (args.options != nullptr) state->options = *static_cast<const cp::ArraySortOptions*>(args.options);
  return state;
}
template <int K>
arrow::Result<std::unique_ptr<cp::KernelState>> SortInitT(cp::KernelContext* c, const cp::KernelInitArgs& a) {
  return SortInitImpl(g_stock_sort[K], c, a);
}

// ArraySortIndices::Exec (vector_array_sort.cc:524-540): output uint64 is preallocated.
Status SortExecImpl(const StockKernel& stock, int key_type, cp::KernelContext* ctx,
                    const cp::ExecSpan& batch, cp::ExecResult* out) {
  auto* state = static_cast<ShimState<cp::ArraySortOptions>*>(ctx->state());
  const ArraySpan& values = batch[0].array;
  if (values.length < g_min_rows.load() || !IsHost(values)) {
    return RunStock(kFnSort, stock, state->stock.get(), ctx, batch, out);
  }
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxSpan dv{};
  const int key_width = (key_type == ARX_KEY_UINT32 || key_type == ARX_KEY_INT32 || key_type == ARX_KEY_FLOAT32) ? 4 : 8;
  ARROW_RETURN_NOT_OK(Upload(values, key_width, kValues, kValidity, st, &dv));
  const int64_t n = values.length;
  const size_t ws_bytes = arx_sort_indices_workspace_bytes(n);
  void *ws = nullptr, *dout = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes, &ws));
  ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, static_cast<size_t>(n) * 8, &dout));
  const int order = state->options.order == cp::SortOrder::Descending ? ARX_SORT_DESCENDING : ARX_SORT_ASCENDING;
  const int placement = state->options.null_placement == cp::NullPlacement::AtStart ? ARX_NULLS_AT_START
                                                                                    : ARX_NULLS_AT_END;
  ARROW_RETURN_NOT_OK(FromArx(arx_sort_indices(&dv, key_type, order, placement, ws, ws_bytes,
                                                  static_cast<uint64_t*>(dout), st)));
  uint64_t* host_out = nullptr;
  if (out->is_array_span()) {
    host_out = out->array_span_mutable()->GetValues<uint64_t>(1);
  } else {
    host_out = out->array_data()->GetMutableValues<uint64_t>(1);
  }
  HIP_RETURN_NOT_OK(hipMemcpyAsync(host_out, dout, static_cast<size_t>(n) * 8, hipMemcpyDeviceToHost, st));
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  CountGpu(kFnSort);
  return Status::OK();
}
// The registered kernels are MemAllocation::NO_PREALLOCATE twins (the stock ones preallocate the
// uint64 output from the CPU pool, vector_array_sort.cc:656-657): device-resident input sorts in
// HBM and the indices stay there; host input gets the buffer the executor would have preallocated
// and runs the preallocated-style exec above.
Status SortExecNP(const StockKernel& stock, int key_type, cp::KernelContext* ctx, const cp::ExecSpan& batch,
                  cp::ExecResult* out) {
  const ArraySpan& values = batch[0].array;
  const int64_t n = values.length;
  ArrayData* out_arr = out->array_data().get();
  out_arr->buffers.resize(2);
  out_arr->buffers[0] = nullptr;
  out_arr->length = n;
  out_arr->null_count = 0;
  if (OnRocm(values)) {
    auto* state = static_cast<ShimState<cp::ArraySortOptions>*>(ctx->state());
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ArxSpan dv{};
    ARROW_RETURN_NOT_OK(DeviceSpan(values, &dv));
    const size_t ws_bytes = arx_sort_indices_workspace_bytes(n);
    void* ws = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes, &ws));
    ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], AllocDevice(n * 8));
    const int order = state->options.order == cp::SortOrder::Descending ? ARX_SORT_DESCENDING : ARX_SORT_ASCENDING;
    const int placement = state->options.null_placement == cp::NullPlacement::AtStart ? ARX_NULLS_AT_START
                                                                                      : ARX_NULLS_AT_END;
    ARROW_RETURN_NOT_OK(FromArx(arx_sort_indices(&dv, key_type, order, placement, ws, ws_bytes,
                                                    reinterpret_cast<uint64_t*>(out_arr->buffers[1]->mutable_address()),
                                                    st)));
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnSort);
    return Status::OK();
  }
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> data, ctx->Allocate(n * 8));
  // the VectorExecutor hands vector kernels an ArrayData (exec.cc:1103-1111): same shape here
  cp::ExecResult tmp;
  tmp.value = ArrayData::Make(out_arr->type, n, {nullptr, data}, /*null_count=*/0);
  ARROW_RETURN_NOT_OK(SortExecImpl(stock, key_type, ctx, batch, &tmp));
  out_arr->buffers[1] = std::move(data);
  return Status::OK();
}
// Chunked input goes to the stock exec_chunked (ArraySortIndicesChunked), which expects the
// preallocated output our NO_PREALLOCATE twin no longer gets from the executor: allocate it here.
Status SortChunkedNP(const StockKernel& stock, cp::KernelContext* ctx, const cp::ExecBatch& batch, arrow::Datum* out) {
  auto* state = static_cast<ShimState<cp::ArraySortOptions>*>(ctx->state());
  ArrayData* out_arr = out->mutable_array();
  out_arr->buffers.resize(2);
  out_arr->buffers[0] = nullptr;
  ARROW_ASSIGN_OR_RAISE(out_arr->buffers[1], ctx->Allocate(batch.length * 8));
  out_arr->null_count = 0;
  CountStock(kFnSort);
  ctx->SetState(state->stock.get());
  const Status st = stock.exec_chunked(ctx, batch, out);
  ctx->SetState(state);
  return st;
}
template <int K>
Status SortChunkedT(cp::KernelContext* c, const cp::ExecBatch& b, arrow::Datum* o) {
  return SortChunkedNP(g_stock_sort[K], c, b, o);
}
template <int K>
Status SortExecT(cp::KernelContext* c, const cp::ExecSpan& b, cp::ExecResult* o) {
  return SortExecNP(g_stock_sort[K], kSortSlotKey[K], c, b, o);
}

// ---------------------------------------------------------------- cast(float64 -> float32)
// Cast kernels live in a private table (GetCastFunction, cpp/src/arrow/compute/cast.cc:207-214)
// whose DispatchExact returns the FIRST exact-type match (cast.cc:170-205), so an added kernel
// would never be chosen.  The public route is the one the registry offers: re-register the
// "cast" MetaFunction (AddFunction(..., allow_overwrite=true), registry.h:69) with a wrapper that
// takes float64 -> float32 arrays and hands every other cast to the stock meta-function.
// Semantics: CastPrimitive<FloatType,DoubleType>::Exec (scalar_cast_internal.cc:41-53): every slot
// converted; validity shared or copied like NullHandling::INTERSECTION does for one input.
class RocmCastMetaFunction : public cp::MetaFunction {
 public:
  explicit RocmCastMetaFunction(std::shared_ptr<cp::Function> stock)
      : cp::MetaFunction("cast", cp::Arity::Unary(), stock->doc(), stock->default_options()),
        stock_(std::move(stock)) {}

  arrow::Result<arrow::Datum> ExecuteImpl(const std::vector<arrow::Datum>& args,
                                          const cp::FunctionOptions* options,
                                          cp::ExecContext* ctx) const override {
    const auto* cast_options = static_cast<const cp::CastOptions*>(options);
    if (cast_options != nullptr && cast_options->to_type.type != nullptr &&
        cast_options->to_type.id() == Type::FLOAT && args.size() == 1 && args[0].is_array() &&
        args[0].array()->type->id() == Type::DOUBLE) {
      ArraySpan in(*args[0].array());
      if (OnRocm(in)) return CastF64F32Device(*args[0].array());
      if (args[0].length() >= g_min_rows_streaming.load() && IsHost(in)) return CastF64F32(*args[0].array(), ctx);
    }
    // integer casts on device-resident arrays: int64 -> int32 (IntegersCanFit unless allow_int_overflow)
    // and int32 -> int64; host arrays stay on the stock kernels
    if (cast_options != nullptr && cast_options->to_type.type != nullptr && args.size() == 1 && args[0].is_array()) {
      const Type::type from = args[0].array()->type->id();
      const Type::type to = cast_options->to_type.id();
      if (((from == Type::INT64 && (to == Type::INT32 || to == Type::DOUBLE)) || (from == Type::INT32 && to == Type::INT64)) &&
          OnRocm(ArraySpan(*args[0].array()))) {
        return CastIntegerDevice(*args[0].array(), to,
                                 to == Type::DOUBLE ? cast_options->allow_float_truncate : cast_options->allow_int_overflow);
      }
    }
    CountStock(kFnCast);
    return stock_->Execute(args, options, ctx);
  }

 private:
  static arrow::Result<arrow::Datum> CastIntegerDevice(const ArrayData& in, Type::type to, bool unchecked) {
    const int64_t n = in.length;
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    const int out_width = to == Type::INT32 ? 4 : 8;
    ARROW_ASSIGN_OR_RAISE(auto out_values, AllocDevice(n * out_width));
    ArxSpan sp{};
    ARROW_RETURN_NOT_OK(DeviceSpan(ArraySpan(in), &sp));
    if (to == Type::INT32) {
      void* ws = nullptr;
      ARROW_RETURN_NOT_OK(t_scratch.Get(kFlag, 64, &ws));
      ARROW_RETURN_NOT_OK(FromArx(arx_cast_i64_i32(&sp, unchecked ? 1 : 0, ws, 64,
                                                   reinterpret_cast<int32_t*>(out_values->mutable_address()), st)));
    } else if (to == Type::DOUBLE) {
      void* ws = nullptr;
      ARROW_RETURN_NOT_OK(t_scratch.Get(kFlag, 64, &ws));
      ARROW_RETURN_NOT_OK(FromArx(arx_cast_i64_f64(&sp, unchecked ? 1 : 0, ws, 64,
                                                   reinterpret_cast<double*>(out_values->mutable_address()), st)));
    } else {
      ARROW_RETURN_NOT_OK(FromArx(arx_cast_i32_i64(static_cast<const int32_t*>(sp.data) + sp.offset, n,
                                                   reinterpret_cast<int64_t*>(out_values->mutable_address()), st)));
    }
    std::shared_ptr<Buffer> validity;
    int64_t null_count = 0;
    if (sp.validity != nullptr && n > 0) {
      if (in.offset == 0) {
        validity = in.buffers[0];
      } else {
        ARROW_ASSIGN_OR_RAISE(validity, AllocDevice(((n + 63) / 64) * 8));
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_copy(sp.validity, sp.offset, n,
                                                    reinterpret_cast<void*>(validity->mutable_address()), st)));
      }
      null_count = in.null_count.load();
      if (null_count < 0 || in.offset != 0) {
        ARROW_ASSIGN_OR_RAISE(null_count, DeviceNullCount(*validity, n, st));
      }
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnCast);
    auto type = to == Type::INT32 ? arrow::int32() : (to == Type::DOUBLE ? arrow::float64() : arrow::int64());
    return arrow::Datum(ArrayData::Make(std::move(type), n, {std::move(validity), std::move(out_values)}, null_count));
  }

  // device-resident input: output values (and a re-based validity bitmap if offset != 0) in HBM
  static arrow::Result<arrow::Datum> CastF64F32Device(const ArrayData& in) {
    const int64_t n = in.length;
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ARROW_ASSIGN_OR_RAISE(auto out_values, AllocDevice(n * 4));
    const double* src = reinterpret_cast<const double*>(in.buffers[1]->address()) + in.offset;
    ARROW_RETURN_NOT_OK(FromArx(arx_cast_f64_f32(src, n, reinterpret_cast<float*>(out_values->mutable_address()), st)));
    std::shared_ptr<Buffer> validity;
    if (in.buffers[0] != nullptr && in.null_count != 0) {
      if (in.offset == 0) {
        validity = in.buffers[0];
      } else {
        ARROW_ASSIGN_OR_RAISE(validity, AllocDevice(((n + 63) / 64) * 8));
        ARROW_RETURN_NOT_OK(FromArx(arx_bitmap_copy(reinterpret_cast<const void*>(in.buffers[0]->address()), in.offset,
                                                    n, reinterpret_cast<void*>(validity->mutable_address()), st)));
      }
    }
    int64_t null_count = 0;
    if (validity) {
      null_count = in.null_count.load();
      if (null_count < 0) {
        ARROW_ASSIGN_OR_RAISE(null_count, DeviceNullCount(*validity, n, st));
      }
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnCast);
    return arrow::Datum(ArrayData::Make(arrow::float32(), n, {std::move(validity), std::move(out_values)}, null_count));
  }

  static arrow::Result<arrow::Datum> CastF64F32(const ArrayData& in, cp::ExecContext* ctx) {
    const int64_t n = in.length;
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    void *din = nullptr, *dout = nullptr;
    ARROW_RETURN_NOT_OK(t_scratch.Get(kValues, static_cast<size_t>(n) * 8 + 16, &din));
    ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, static_cast<size_t>(n) * 4 + 16, &dout));
    ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> out_values,
                          arrow::AllocateBuffer(n * 4, ctx->memory_pool()));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(din, in.GetValues<double>(1), static_cast<size_t>(n) * 8,
                                     hipMemcpyHostToDevice, st));
    ARROW_RETURN_NOT_OK(FromArx(arx_cast_f64_f32(static_cast<const double*>(din), n,
                                                 static_cast<float*>(dout), st)));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(out_values->mutable_data(), dout, static_cast<size_t>(n) * 4,
                                     hipMemcpyDeviceToHost, st));
    std::shared_ptr<Buffer> validity;
    if (in.null_count != 0 && in.buffers[0] != nullptr) {
      if (in.offset == 0) {
        validity = in.buffers[0];  // zero-copy, as PropagateNulls does (exec.cc:1222-1281)
      } else {
        ARROW_ASSIGN_OR_RAISE(validity, arrow::internal::CopyBitmap(ctx->memory_pool(),
                                                                    in.buffers[0]->data(),
                                                                    in.offset, n));
      }
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    CountGpu(kFnCast);
    const int64_t null_count = validity ? in.null_count.load() : 0;
    return arrow::Datum(ArrayData::Make(arrow::float32(), n, {std::move(validity), std::move(out_values)},
                                        null_count));
  }

  std::shared_ptr<cp::Function> stock_;
};

// ---------------------------------------------------------------- hash_sum(int64, uint32)
// The HashAggregateKernel vtable (compute/kernel.h:720-769) of
// GroupedReducingAggregator<Int64Type,GroupedSumImpl> (hash_aggregate_numeric.cc:44-187) with the
// per-group state in HBM.  Acero's GroupByNode drives it unchanged
// (acero/groupby_aggregate_node.cc:210-337): resize after every Grouper::Consume, consume with
// the dense uint32 group ids, merge thread-local states through a group_id_mapping, finalize.
struct DeviceSumState : public cp::KernelState {
  cp::ScalarAggregateOptions options;
  int64_t num_groups = 0;
  int64_t capacity = 0;
  int64_t* sums = nullptr;
  int64_t* counts = nullptr;
  uint32_t* null_seen = nullptr;
  ~DeviceSumState() override {
    if (sums) (void)hipFree(sums);
    if (counts) (void)hipFree(counts);
    if (null_seen) (void)hipFree(null_seen);
  }
};

arrow::Result<std::unique_ptr<cp::KernelState>> HashSumInit(cp::KernelContext*,
                                                            const cp::KernelInitArgs& args) {
  auto state = std::make_unique<DeviceSumState>();
  if (args.options != nullptr) {
    state->options = *static_cast<const cp::ScalarAggregateOptions*>(args.options);
  }
  return state;
}

template <typename T>
Status GrowDevice(T** ptr, int64_t old_n, int64_t new_cap, hipStream_t st) {
  T* fresh = nullptr;
  HIP_RETURN_NOT_OK(hipMalloc(reinterpret_cast<void**>(&fresh), static_cast<size_t>(new_cap) * sizeof(T)));
  HIP_RETURN_NOT_OK(hipMemsetAsync(fresh, 0, static_cast<size_t>(new_cap) * sizeof(T), st));
  if (old_n > 0) {
    HIP_RETURN_NOT_OK(hipMemcpyAsync(fresh, *ptr, static_cast<size_t>(old_n) * sizeof(T),
                                     hipMemcpyDeviceToDevice, st));
  }
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  if (*ptr) HIP_RETURN_NOT_OK(hipFree(*ptr));
  *ptr = fresh;
  return Status::OK();
}

// Resize (:61-68): new groups start at sum 0 / count 0 / no null seen.
Status HashSumResize(cp::KernelContext* ctx, int64_t new_num_groups) {
  auto* s = static_cast<DeviceSumState*>(ctx->state());
  if (new_num_groups > s->capacity) {
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    const int64_t cap = std::max<int64_t>({new_num_groups, 2 * s->capacity, 1024});
    ARROW_RETURN_NOT_OK(GrowDevice(&s->sums, s->num_groups, cap, st));
    ARROW_RETURN_NOT_OK(GrowDevice(&s->counts, s->num_groups, cap, st));
    ARROW_RETURN_NOT_OK(GrowDevice(&s->null_seen, s->num_groups, cap, st));
    s->capacity = cap;
  }
  s->num_groups = new_num_groups;
  return Status::OK();
}

// Consume (:70-83): batch = {values (array | scalar), group ids (uint32 array)}.
Status HashSumConsume(cp::KernelContext* ctx, const cp::ExecSpan& batch) {
  auto* s = static_cast<DeviceSumState*>(ctx->state());
  const ArraySpan& gids = batch[1].array;
  const int64_t n = gids.length;
  if (n == 0) return Status::OK();
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  ArxSpan dg{};
  ARROW_RETURN_NOT_OK(Upload(gids, 4, kArg2, kArg2Validity, st, &dg));
  const uint32_t* d_gids = static_cast<const uint32_t*>(dg.data) + dg.offset;
  ArxSpan dv{};
  int is_scalar = 0;
  int64_t scalar_value = 0;
  if (batch[0].is_array()) {
    ARROW_RETURN_NOT_OK(Upload(batch[0].array, 8, kValues, kValidity, st, &dv));
  } else {
    const arrow::Scalar& sc = *batch[0].scalar;
    is_scalar = 1;
    dv.length = n;
    dv.null_count = sc.is_valid ? 0 : n;
    if (sc.is_valid) scalar_value = static_cast<const arrow::Int64Scalar&>(sc).value;
  }
  ARROW_RETURN_NOT_OK(FromArx(arx_hash_sum_i64_consume(&dv, is_scalar, scalar_value, d_gids, n, s->sums,
                                                       s->counts, s->null_seen, st)));
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  CountGpu(kFnHashSum);
  return Status::OK();
}

// Merge (:85-107).
Status HashSumMerge(cp::KernelContext* ctx, cp::KernelState&& other_state, const ArrayData& mapping) {
  auto* s = static_cast<DeviceSumState*>(ctx->state());
  auto* other = static_cast<DeviceSumState*>(&other_state);
  const int64_t g = mapping.length;
  if (g == 0) return Status::OK();
  hipStream_t st;
  ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
  void* d_map = nullptr;
  ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2, static_cast<size_t>(g) * 4 + 16, &d_map));
  HIP_RETURN_NOT_OK(hipMemcpyAsync(d_map, mapping.GetValues<uint32_t>(1), static_cast<size_t>(g) * 4,
                                   hipMemcpyHostToDevice, st));
  ARROW_RETURN_NOT_OK(FromArx(arx_hash_sum_i64_merge(s->sums, s->counts, s->null_seen, other->sums,
                                                     other->counts, other->null_seen,
                                                     static_cast<const uint32_t*>(d_map), g, st)));
  HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
  return Status::OK();
}

// Finalize (:130-152) + Finish (:109-128).
Status HashSumFinalize(cp::KernelContext* ctx, arrow::Datum* out) {
  auto* s = static_cast<DeviceSumState*>(ctx->state());
  const int64_t g = s->num_groups;
  arrow::MemoryPool* pool = ctx->memory_pool();
  ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> values, arrow::AllocateBuffer(g * 8, pool));
  std::shared_ptr<Buffer> bitmap;
  int64_t null_count = 0;
  if (g > 0) {
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    void *d_bits = nullptr, *d_counter = nullptr;
    const size_t words = static_cast<size_t>((g + 63) / 64);
    ARROW_RETURN_NOT_OK(t_scratch.Get(kOutValidity, words * 8 + 16, &d_bits));
    ARROW_RETURN_NOT_OK(t_scratch.Get(kCounter, 64, &d_counter));
    HIP_RETURN_NOT_OK(hipMemsetAsync(d_counter, 0, 8, st));
    ARROW_RETURN_NOT_OK(FromArx(arx_hash_sum_i64_finalize(s->counts, s->null_seen, g,
                                                          s->options.skip_nulls ? 1 : 0,
                                                          s->options.min_count, d_bits,
                                                          static_cast<int64_t*>(d_counter), st)));
    ARROW_ASSIGN_OR_RAISE(bitmap, arrow::AllocateBitmap(g, pool));
    int64_t valid_count = 0;
    HIP_RETURN_NOT_OK(hipMemcpyAsync(values->mutable_data(), s->sums, static_cast<size_t>(g) * 8,
                                     hipMemcpyDeviceToHost, st));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(bitmap->mutable_data(), d_bits,
                                     static_cast<size_t>(arrow::bit_util::BytesForBits(g)),
                                     hipMemcpyDeviceToHost, st));
    HIP_RETURN_NOT_OK(hipMemcpyAsync(&valid_count, d_counter, 8, hipMemcpyDeviceToHost, st));
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    null_count = g - valid_count;
    if (s->options.skip_nulls) {
      if (null_count == 0) bitmap = nullptr;  // Finish allocates a bitmap only when a group is null
    } else {
      null_count = arrow::kUnknownNullCount;
    }
  }
  *out = arrow::Datum(ArrayData::Make(arrow::int64(), g, {std::move(bitmap), std::move(values)}, null_count));
  return Status::OK();
}

// ---------------------------------------------------------------- Acero: fused group-by node
// Whole-operator replacement: an ExecNode registered with
// default_exec_factory_registry()->AddFactory("aggregate_rocm", ...) (acero/exec_plan.h:353-373;
// names must be new, exec_plan.cc:1132-1142) that takes the place of GroupByNode
// (acero/groupby_aggregate_node.cc:210-337) for `hash_sum(int64) GROUP BY int32`: instead of a CPU
// Grouper feeding dense ids to the aggregate kernel, keys and values go to the fused device
// operator (arx_groupby_*: hash partition -> LDS tables -> HBM table).  Batches without nulls are
// appended to device staging buffers and consumed in ONE partitioned pass at InputFinished; a batch
// with nulls is consumed on arrival.  Output = key column ++ aggregate column like
// GroupByNode::Finalize (:300-333), row order unspecified (as the reference with threads).
namespace ac = arrow::acero;

class RocmGroupBySumNode : public ac::ExecNode {
 public:
  // one output column per requested aggregate; all of them read the same fused per-group state
  // (wrap-around sum, count of valid values, "a null value was seen")
  enum AggKind { kSum, kCount, kMin, kMax };
  struct AggSpec {
    AggKind kind = kSum;                 // hash_sum | hash_count (ONLY_VALID) | hash_min | hash_max
    cp::ScalarAggregateOptions options;  // sum: skip_nulls / min_count; min/max: skip_nulls
  };

  RocmGroupBySumNode(ac::ExecPlan* plan, std::vector<ac::ExecNode*> inputs,
                     std::shared_ptr<arrow::Schema> out_schema, int key_idx, int val_idx,
                     std::vector<AggSpec> aggs)
      : ac::ExecNode(plan, std::move(inputs), {"input"}, std::move(out_schema)),
        key_idx_(key_idx), val_idx_(val_idx), aggs_(std::move(aggs)) {
    for (const auto& a : aggs_) needs_minmax_ = needs_minmax_ || a.kind == kMin || a.kind == kMax;
  }

  ~RocmGroupBySumNode() override {
    for (void* p : {state_, minmax_, d_keys_, d_vals_}) {
      if (p) (void)hipFree(p);
    }
  }

  static arrow::Result<ac::ExecNode*> Make(ac::ExecPlan* plan, std::vector<ac::ExecNode*> inputs,
                                           const ac::ExecNodeOptions& options) {
    if (inputs.size() != 1) return Status::Invalid("aggregate_rocm takes exactly one input");
    const auto* opts = dynamic_cast<const ac::AggregateNodeOptions*>(&options);
    if (opts == nullptr) return Status::TypeError("aggregate_rocm expects AggregateNodeOptions");
    if (opts->keys.size() != 1 || !opts->segment_keys.empty() || opts->aggregates.empty()) {
      return Status::NotImplemented("aggregate_rocm: one key, no segment keys, at least one aggregate");
    }
    const auto& in_schema = *inputs[0]->output_schema();
    ARROW_ASSIGN_OR_RAISE(auto kpath, opts->keys[0].FindOne(in_schema));
    if (kpath.indices().size() != 1) return Status::NotImplemented("aggregate_rocm: nested field references");
    const int ki = kpath[0];
    int vi = -1;
    std::vector<AggSpec> aggs;
    std::vector<std::shared_ptr<arrow::Field>> fields{in_schema.field(ki)};
    for (const auto& agg : opts->aggregates) {
      AggSpec spec;
      if (agg.function == "hash_min" || agg.function == "hash_max") {
        spec.kind = agg.function == "hash_min" ? kMin : kMax;
        if (agg.options != nullptr) {
          const auto* so = dynamic_cast<const cp::ScalarAggregateOptions*>(agg.options.get());
          if (so == nullptr) return Status::TypeError("aggregate_rocm: ", agg.function, " takes ScalarAggregateOptions");
          spec.options = *so;
        }
      } else if (agg.function == "hash_count") {
        spec.kind = kCount;
        if (agg.options != nullptr) {
          const auto* co = dynamic_cast<const cp::CountOptions*>(agg.options.get());
          if (co == nullptr) return Status::TypeError("aggregate_rocm: hash_count takes CountOptions");
          if (co->mode != cp::CountOptions::ONLY_VALID) {
            return Status::NotImplemented("aggregate_rocm: hash_count with CountOptions::ONLY_VALID only");
          }
        }
      } else if (agg.function == "hash_sum") {
        if (agg.options != nullptr) {
          const auto* so = dynamic_cast<const cp::ScalarAggregateOptions*>(agg.options.get());
          if (so == nullptr) return Status::TypeError("aggregate_rocm: hash_sum takes ScalarAggregateOptions");
          spec.options = *so;
        }
      } else {
        return Status::NotImplemented("aggregate_rocm: hash_sum / hash_count / hash_min / hash_max only, got ",
                                      agg.function);
      }
      if (agg.target.size() != 1) return Status::NotImplemented("aggregate_rocm: unary aggregates only");
      ARROW_ASSIGN_OR_RAISE(auto vpath, agg.target[0].FindOne(in_schema));
      if (vpath.indices().size() != 1) return Status::NotImplemented("aggregate_rocm: nested field references");
      if (vi >= 0 && vpath[0] != vi) {
        return Status::NotImplemented("aggregate_rocm: all aggregates must read the same value column");
      }
      vi = vpath[0];
      aggs.push_back(spec);
      fields.push_back(arrow::field(agg.name, arrow::int64()));
    }
    if (in_schema.field(ki)->type()->id() != Type::INT32 || in_schema.field(vi)->type()->id() != Type::INT64) {
      return Status::NotImplemented("aggregate_rocm: hash_sum(int64) GROUP BY int32 only, got key ",
                                    in_schema.field(ki)->type()->ToString(), " value ",
                                    in_schema.field(vi)->type()->ToString());
    }
    return plan->EmplaceNode<RocmGroupBySumNode>(plan, std::move(inputs), arrow::schema(std::move(fields)), ki, vi,
                                                 std::move(aggs));
  }

  const char* kind_name() const override { return "RocmGroupBySumNode"; }

  Status InputReceived(ac::ExecNode*, cp::ExecBatch batch) override {
    {
      std::lock_guard<std::mutex> lock(mu_);
      ARROW_RETURN_NOT_OK(Consume(batch));
    }
    if (counter_.Increment()) return Finish();
    return Status::OK();
  }
  Status InputFinished(ac::ExecNode*, int total_batches) override {
    if (counter_.SetTotal(total_batches)) return Finish();
    return Status::OK();
  }
  Status StartProducing() override { return Status::OK(); }
  void PauseProducing(ac::ExecNode*, int32_t) override {}
  void ResumeProducing(ac::ExecNode*, int32_t) override {}

 protected:
  Status StopProducingImpl() override { return Status::OK(); }

 private:
  static constexpr int64_t kMaxCapacity = int64_t(1) << 28;

  Status EnsureState(int64_t more_rows) {
    // capacity: a power of two above twice the distinct keys possible so far
    const int64_t bound = std::min<int64_t>(kMaxCapacity, 2 * (rows_seen_ + more_rows) + 2);
    int64_t cap = 1 << 16;
    while (cap < bound) cap <<= 1;
    if (state_ != nullptr && cap <= capacity_) return Status::OK();
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    void* fresh = nullptr;
    void* fresh_mm = nullptr;
    HIP_RETURN_NOT_OK(hipMalloc(&fresh, arx_groupby_state_bytes(cap)));
    ARROW_RETURN_NOT_OK(FromArx(arx_groupby_init(fresh, cap, st)));
    if (needs_minmax_) {
      HIP_RETURN_NOT_OK(hipMalloc(&fresh_mm, arx_groupby_minmax_bytes(cap)));
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_init(fresh_mm, cap, st)));
    }
    if (state_ != nullptr) {  // rehash: export the old table's partial aggregates, merge them
      int64_t g = 0;
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_num_groups(state_, &g, st)));
      if (g > 0) {
        void *k, *kv, *s, *c, *nn;
        ARROW_RETURN_NOT_OK(t_scratch.Get(kValues, g * 4 + 16, &k));
        ARROW_RETURN_NOT_OK(t_scratch.Get(kValidity, g + 16, &kv));
        ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2, g * 8 + 16, &s));
        ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2Validity, g * 8 + 16, &c));
        ARROW_RETURN_NOT_OK(t_scratch.Get(kOutValidity, g + 16, &nn));
        void *mn = nullptr, *mx = nullptr;
        if (needs_minmax_) {
          ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, g * 8 + 16, &mn));
          ARROW_RETURN_NOT_OK(t_scratch.Get(kBinWs, g * 8 + 16, &mx));
        }
        ARROW_RETURN_NOT_OK(FromArx(arx_groupby_export(state_, minmax_, (int32_t*)k, (uint8_t*)kv, (int64_t*)s,
                                                       (int64_t*)c, (uint8_t*)nn, (int64_t*)mn, (int64_t*)mx, st)));
        ARROW_RETURN_NOT_OK(FromArx(arx_groupby_sum_i64_merge(fresh, cap, (const int32_t*)k, (const uint8_t*)kv,
                                                              (const int64_t*)s, (const int64_t*)c,
                                                              (const uint8_t*)nn, g, st)));
        if (needs_minmax_) {
          ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_merge(fresh, fresh_mm, cap, (const int32_t*)k,
                                                               (const uint8_t*)kv, (const int64_t*)mn,
                                                               (const int64_t*)mx, (const uint8_t*)nn, g, st)));
        }
      }
      HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
      HIP_RETURN_NOT_OK(hipFree(state_));
      if (minmax_) HIP_RETURN_NOT_OK(hipFree(minmax_));
    }
    state_ = fresh;
    minmax_ = fresh_mm;
    capacity_ = cap;
    return Status::OK();
  }

  Status Consume(const cp::ExecBatch& batch) {
    const int64_t n = batch.length;
    if (n == 0) return Status::OK();
    if (!batch[key_idx_].is_array() || !batch[val_idx_].is_array()) {
      return Status::NotImplemented("aggregate_rocm: scalar columns");
    }
    const ArrayData& k = *batch[key_idx_].array();
    const ArrayData& v = *batch[val_idx_].array();
    // device-resident batches (kROCM buffers): no staging over PCIe, and never a CPU popcount —
    // a sliced device array with an unknown null count is treated as "may have nulls"
    const bool on_device = DataOnRocm(k) || DataOnRocm(v);
    if (on_device && !(DataOnRocm(k) && DataOnRocm(v))) {
      return Status::Invalid("aggregate_rocm: key and value columns must both be host or both be device arrays");
    }
    const bool nulls = on_device ? (k.buffers[0] != nullptr && k.null_count.load() != 0) ||
                                       (v.buffers[0] != nullptr && v.null_count.load() != 0)
                                 : (k.GetNullCount() != 0) || (v.GetNullCount() != 0);
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    if (!nulls) {
      // append to the device staging buffers; consumed in one partitioned pass at the end
      if (staged_ + n > staged_cap_) {
        const int64_t cap = std::max<int64_t>(2 * staged_cap_, std::max<int64_t>(staged_ + n, 1 << 20));
        void *nk = nullptr, *nv = nullptr;
        HIP_RETURN_NOT_OK(hipMalloc(&nk, cap * 4));
        HIP_RETURN_NOT_OK(hipMalloc(&nv, cap * 8));
        if (staged_ > 0) {
          HIP_RETURN_NOT_OK(hipMemcpy(nk, d_keys_, staged_ * 4, hipMemcpyDeviceToDevice));
          HIP_RETURN_NOT_OK(hipMemcpy(nv, d_vals_, staged_ * 8, hipMemcpyDeviceToDevice));
        }
        if (d_keys_) HIP_RETURN_NOT_OK(hipFree(d_keys_));
        if (d_vals_) HIP_RETURN_NOT_OK(hipFree(d_vals_));
        d_keys_ = nk;
        d_vals_ = nv;
        staged_cap_ = cap;
      }
      const void* ksrc = on_device ? reinterpret_cast<const void*>(k.buffers[1]->address() + k.offset * 4)
                                   : static_cast<const void*>(k.GetValues<int32_t>(1));
      const void* vsrc = on_device ? reinterpret_cast<const void*>(v.buffers[1]->address() + v.offset * 8)
                                   : static_cast<const void*>(v.GetValues<int64_t>(1));
      const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
      HIP_RETURN_NOT_OK(hipMemcpy(static_cast<int32_t*>(d_keys_) + staged_, ksrc, n * 4, kind));
      HIP_RETURN_NOT_OK(hipMemcpy(static_cast<int64_t*>(d_vals_) + staged_, vsrc, n * 8, kind));
      staged_ += n;
      return Status::OK();
    }
    if (on_device) {
      ARROW_RETURN_NOT_OK(EnsureState(staged_ + n));
      auto span_of = [](const ArrayData& a) {
        ArxSpan sp{};
        sp.validity = a.buffers[0] ? reinterpret_cast<const void*>(a.buffers[0]->address()) : nullptr;
        sp.data = reinterpret_cast<const void*>(a.buffers[1]->address());
        sp.offset = a.offset;
        sp.length = a.length;
        const int64_t nc = a.null_count.load();
        sp.null_count = sp.validity == nullptr ? 0 : (nc > 0 ? nc : arrow::kUnknownNullCount);
        return sp;
      };
      ArxSpan dk = span_of(k), dv = span_of(v);
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_sum_i64_consume(state_, capacity_, &dk, &dv, nullptr, 0, st)));
      if (needs_minmax_) {
        ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_i64_consume(state_, minmax_, capacity_, &dk, &dv, st)));
      }
      HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
      rows_seen_ += n;
      CountGpu(kFnHashSum);
      return Status::OK();
    }
    ARROW_RETURN_NOT_OK(EnsureState(staged_ + n));
    ArxSpan dk{}, dv{};
    ARROW_RETURN_NOT_OK(Upload(ArraySpan(k), 4, kValues, kValidity, st, &dk));
    ARROW_RETURN_NOT_OK(Upload(ArraySpan(v), 8, kArg2, kArg2Validity, st, &dv));
    ARROW_RETURN_NOT_OK(FromArx(arx_groupby_sum_i64_consume(state_, capacity_, &dk, &dv, nullptr, 0, st)));
    if (needs_minmax_) {
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_i64_consume(state_, minmax_, capacity_, &dk, &dv, st)));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    rows_seen_ += n;
    CountGpu(kFnHashSum);
    return Status::OK();
  }

  Status Finish() {
    std::lock_guard<std::mutex> lock(mu_);
    hipStream_t st;
    ARROW_RETURN_NOT_OK(t_scratch.Stream(&st));
    ARROW_RETURN_NOT_OK(EnsureState(staged_));
    if (staged_ > 0) {
      ArxSpan dk{nullptr, d_keys_, 0, staged_, 0}, dv{nullptr, d_vals_, 0, staged_, 0};
      const size_t ws_bytes = arx_groupby_consume_workspace_bytes(staged_, capacity_);
      void* ws = nullptr;
      if (ws_bytes > 0) {
        ARROW_RETURN_NOT_OK(t_scratch.Get(kWs, ws_bytes + 256, &ws));
        ws = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
      }
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_sum_i64_consume(state_, capacity_, &dk, &dv, ws, ws_bytes, st)));
      if (needs_minmax_) {
        ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_i64_consume(state_, minmax_, capacity_, &dk, &dv, st)));
      }
      rows_seen_ += staged_;
      staged_ = 0;
      CountGpu(kFnHashSum);
    }
    int64_t g = 0;
    ARROW_RETURN_NOT_OK(FromArx(arx_groupby_num_groups(state_, &g, st)));
    arrow::MemoryPool* pool = plan_->query_context()->memory_pool();
    ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> keys, arrow::AllocateBuffer(g * 4, pool));
    std::vector<uint8_t> key_valid(g);
    std::vector<arrow::Datum> columns(1 + aggs_.size());
    void *k = nullptr, *kv = nullptr, *s = nullptr, *c = nullptr, *nn = nullptr, *ok = nullptr;
    void *mn = nullptr, *mx = nullptr;
    if (g > 0) {
      ARROW_RETURN_NOT_OK(t_scratch.Get(kValues, g * 4 + 16, &k));
      ARROW_RETURN_NOT_OK(t_scratch.Get(kValidity, g + 16, &kv));
      ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2, g * 8 + 16, &s));
      ARROW_RETURN_NOT_OK(t_scratch.Get(kArg2Validity, g * 8 + 16, &c));
      ARROW_RETURN_NOT_OK(t_scratch.Get(kOutValidity, g + 16, &nn));
      ARROW_RETURN_NOT_OK(t_scratch.Get(kOutData, g + 16, &ok));
      if (needs_minmax_) {
        ARROW_RETURN_NOT_OK(t_scratch.Get(kBinWs, g * 16 + 32, &mn));
        mx = static_cast<uint8_t*>(mn) + ((g * 8 + 15) & ~int64_t(15));
      }
      ARROW_RETURN_NOT_OK(FromArx(arx_groupby_export(state_, minmax_, (int32_t*)k, (uint8_t*)kv, (int64_t*)s,
                                                     (int64_t*)c, (uint8_t*)nn, (int64_t*)mn, (int64_t*)mx, st)));
      HIP_RETURN_NOT_OK(hipMemcpyAsync(keys->mutable_data(), k, g * 4, hipMemcpyDeviceToHost, st));
      HIP_RETURN_NOT_OK(hipMemcpyAsync(key_valid.data(), kv, g, hipMemcpyDeviceToHost, st));
    }
    for (size_t ai = 0; ai < aggs_.size(); ++ai) {
      const AggSpec& spec = aggs_[ai];
      ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> data, arrow::AllocateBuffer(g * 8, pool));
      std::shared_ptr<Buffer> bits;
      if (spec.kind == kCount) {
        // GroupedCountImpl (hash_aggregate.cc): the count of valid values, never null
        if (g > 0) HIP_RETURN_NOT_OK(hipMemcpyAsync(data->mutable_data(), c, g * 8, hipMemcpyDeviceToHost, st));
        HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
      } else if (spec.kind == kMin || spec.kind == kMax) {
        // GroupedMinMaxImpl::Finalize (hash_aggregate.cc:401-419): null where the group saw no value
        // (or, with !skip_nulls, saw a null); the reference writes the anti-extremum there, we write 0
        // only into the validity — the data slot keeps the anti-extremum as well
        std::vector<uint8_t> mm_valid(g);
        if (g > 0) {
          ARROW_RETURN_NOT_OK(FromArx(arx_groupby_minmax_finalize((const int64_t*)mn, (const int64_t*)mx,
                                                                  (const uint8_t*)nn, g,
                                                                  spec.options.skip_nulls ? 1 : 0, (uint8_t*)ok, st)));
          HIP_RETURN_NOT_OK(hipMemcpyAsync(data->mutable_data(), spec.kind == kMin ? mn : mx, g * 8,
                                           hipMemcpyDeviceToHost, st));
          HIP_RETURN_NOT_OK(hipMemcpyAsync(mm_valid.data(), ok, g, hipMemcpyDeviceToHost, st));
        }
        HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
        ARROW_ASSIGN_OR_RAISE(bits, arrow::internal::BytesToBits(mm_valid, pool));
      } else {
        std::vector<uint8_t> sum_valid(g);
        if (g > 0) {
          ARROW_RETURN_NOT_OK(FromArx(arx_groupby_sum_i64_finalize((const int64_t*)c, (const uint8_t*)nn, g,
                                                                   spec.options.skip_nulls ? 1 : 0,
                                                                   spec.options.min_count, (uint8_t*)ok, st)));
          HIP_RETURN_NOT_OK(hipMemcpyAsync(data->mutable_data(), s, g * 8, hipMemcpyDeviceToHost, st));
          HIP_RETURN_NOT_OK(hipMemcpyAsync(sum_valid.data(), ok, g, hipMemcpyDeviceToHost, st));
        }
        HIP_RETURN_NOT_OK(hipStreamSynchronize(st));  // `ok` is reused by the next aggregate
        ARROW_ASSIGN_OR_RAISE(bits, arrow::internal::BytesToBits(sum_valid, pool));
      }
      std::vector<std::shared_ptr<Buffer>> bufs{std::move(bits), std::move(data)};
      columns[1 + ai] = arrow::Datum(ArrayData::Make(arrow::int64(), g, std::move(bufs)));
    }
    HIP_RETURN_NOT_OK(hipStreamSynchronize(st));
    ARROW_ASSIGN_OR_RAISE(std::shared_ptr<Buffer> kbits, arrow::internal::BytesToBits(key_valid, pool));
    std::vector<std::shared_ptr<Buffer>> kbufs{std::move(kbits), std::move(keys)};
    columns[0] = arrow::Datum(ArrayData::Make(arrow::int32(), g, std::move(kbufs)));
    cp::ExecBatch out(std::move(columns), g);
    const int64_t batch_size = 32768;
    const int nb = static_cast<int>(std::max<int64_t>(1, (g + batch_size - 1) / batch_size));
    for (int i = 0; i < nb; ++i) {
      ARROW_RETURN_NOT_OK(output_->InputReceived(this, out.Slice(i * batch_size, batch_size)));
    }
    return output_->InputFinished(this, nb);
  }

  const int key_idx_, val_idx_;
  const std::vector<AggSpec> aggs_;
  bool needs_minmax_ = false;
  void* minmax_ = nullptr;   // mins | maxes per slot (arx_groupby_minmax_bytes), only with hash_min / hash_max
  std::mutex mu_;
  ac::AtomicCounter counter_;
  void* state_ = nullptr;
  int64_t capacity_ = 0;
  void* d_keys_ = nullptr;
  void* d_vals_ = nullptr;
  int64_t staged_ = 0, staged_cap_ = 0, rows_seen_ = 0;
};

// ---------------------------------------------------------------- registration
// A value type of array_filter / array_take: the concrete type used to find the stock kernel and
// the matcher the added kernel is registered under (parametric types match by type id).
struct ValueType {
  std::shared_ptr<arrow::DataType> probe;
  cp::InputType match;
  ValueType(std::shared_ptr<arrow::DataType> t) : probe(t), match(t) {}  // NOLINT
  ValueType(std::shared_ptr<arrow::DataType> t, Type::type id) : probe(std::move(t)), match(id) {}
};

std::vector<ValueType> FilterValueTypes() {
  return {arrow::int8(), arrow::uint8(), arrow::int16(), arrow::uint16(), arrow::int32(), arrow::uint32(),
          arrow::int64(), arrow::uint64(), arrow::float16(), arrow::float32(), arrow::float64(), arrow::date32(),
          arrow::date64(), arrow::month_interval(), arrow::day_time_interval(), arrow::month_day_nano_interval(),
          {arrow::time32(arrow::TimeUnit::SECOND), Type::TIME32},
          {arrow::time64(arrow::TimeUnit::NANO), Type::TIME64},
          {arrow::timestamp(arrow::TimeUnit::NANO), Type::TIMESTAMP},
          {arrow::duration(arrow::TimeUnit::NANO), Type::DURATION},
          {arrow::decimal128(38, 9), Type::DECIMAL128},
          {arrow::fixed_size_binary(16), Type::FIXED_SIZE_BINARY}};
}

Status RegisterVector(cp::FunctionRegistry* reg, const std::string& name,
                      const std::vector<ValueType>& first_types,
                      const std::vector<cp::InputType>& second, cp::KernelInit init,
                      cp::ArrayKernelExec exec, StockKernel* stock,
                      cp::VectorKernel::ChunkedExec chunked = nullptr) {
  ARROW_ASSIGN_OR_RAISE(auto fn, reg->GetFunction(name));
  if (fn->kind() != cp::Function::VECTOR) return Status::Invalid(name, " is not a vector function");
  auto* vfn = static_cast<cp::VectorFunction*>(fn.get());
  for (const auto& vt : first_types) {
    std::vector<arrow::TypeHolder> probe{vt.probe};
    std::vector<cp::InputType> in{vt.match};
    if (!second.empty()) {
      // probe with a concrete second argument type
      probe.push_back(name == "array_filter" ? arrow::boolean() : arrow::int32());
    }
    ARROW_ASSIGN_OR_RAISE(const cp::Kernel* k0, vfn->DispatchExact(probe));
    cp::VectorKernel copy = *static_cast<const cp::VectorKernel*>(k0);
    if (stock->exec == nullptr) {
      stock->exec = copy.exec;
      stock->init = copy.init;
      stock->exec_chunked = copy.exec_chunked;
    } else if (stock->exec != copy.exec) {
      continue;  // a different stock kernel handles this type: leave it alone
    }
    for (const auto& s : second) in.push_back(s);
    copy.signature = cp::KernelSignature::Make(in, copy.signature->out_type());
    copy.init = init;
    copy.exec = exec;
    if (name == "array_sort_indices") {
      copy.mem_allocation = cp::MemAllocation::NO_PREALLOCATE;
      if (copy.exec_chunked != nullptr) copy.exec_chunked = chunked;
    }
    ARROW_RETURN_NOT_OK(vfn->AddKernel(std::move(copy)));
  }
  return Status::OK();
}

Status RegisterAll() {
  ARROW_RETURN_NOT_OK(cp::Initialize());
  const int ndev = arx_device_count();
  // ARROW_AMD_PLUGIN_DRY_RUN=1: register without a device so the registration logic itself can be
  // exercised on a GPU-less box (tests/test_plugin_registration.py); any call large enough to be
  // routed to the HIP kernels then fails loudly with a HIP error — nothing computes on the CPU here.
  const char* dry = std::getenv("ARROW_AMD_PLUGIN_DRY_RUN");
  if (ndev < 1 && !(dry != nullptr && dry[0] == '1')) {
    return Status::Invalid("arrow_amd: no HIP device visible (", arx_last_error(),
                           "); nothing registered, Arrow keeps its stock kernels");
  }
  // arrays can now be imported onto the MI355X through the C Device Data interface
  {
    const Status st = arrow::RegisterDeviceMapper(arrow::DeviceAllocationType::kROCM, RocmMemoryManagerFor);
    if (!st.ok() && !st.IsKeyError()) return st;  // KeyError: somebody registered kROCM before us
  }
  cp::FunctionRegistry* reg = cp::GetFunctionRegistry();
  ARROW_RETURN_NOT_OK(RegisterVector(reg, "array_filter", FilterValueTypes(), {cp::InputType(arrow::boolean())},
                                     FilterInit, FilterExec, &g_stock_filter));
  ARROW_RETURN_NOT_OK(RegisterVector(reg, "array_take", FilterValueTypes(),
                                     {cp::InputType(cp::match::Integer())}, TakeInit, TakeExec,
                                     &g_stock_take));
  ARROW_RETURN_NOT_OK(RegisterVector(reg, "array_filter", {arrow::utf8(), arrow::binary()},
                                     {cp::InputType(arrow::boolean())}, BinaryFilterInit, BinaryFilterExec,
                                     &g_stock_filter_bin));
  ARROW_RETURN_NOT_OK(RegisterVector(reg, "array_take", {arrow::utf8(), arrow::binary()},
                                     {cp::InputType(cp::match::Integer())}, BinaryTakeInit, BinaryTakeExec,
                                     &g_stock_take_bin));
#define ARX_REGISTER_SORT(K, TYPE)                                                                           \
  ARROW_RETURN_NOT_OK(RegisterVector(reg, "array_sort_indices", {TYPE}, {}, SortInitT<K>, SortExecT<K>,         \
                                     &g_stock_sort[K], SortChunkedT<K>))
  ARX_REGISTER_SORT(ARX_KEY_UINT64, arrow::uint64());
  ARX_REGISTER_SORT(ARX_KEY_INT64, arrow::int64());
  ARX_REGISTER_SORT(ARX_KEY_UINT32, arrow::uint32());
  ARX_REGISTER_SORT(ARX_KEY_INT32, arrow::int32());
  ARX_REGISTER_SORT(ARX_KEY_FLOAT64, arrow::float64());
  ARX_REGISTER_SORT(ARX_KEY_FLOAT32, arrow::float32());
  ARX_REGISTER_SORT(6, arrow::date32());
  ARX_REGISTER_SORT(7, arrow::date64());
  ARX_REGISTER_SORT(8, ValueType(arrow::timestamp(arrow::TimeUnit::NANO), Type::TIMESTAMP));
  ARX_REGISTER_SORT(9, ValueType(arrow::duration(arrow::TimeUnit::NANO), Type::DURATION));
  ARX_REGISTER_SORT(10, ValueType(arrow::time32(arrow::TimeUnit::SECOND), Type::TIME32));
  ARX_REGISTER_SORT(11, ValueType(arrow::time64(arrow::TimeUnit::NANO), Type::TIME64));
#undef ARX_REGISTER_SORT
  {
    ARROW_ASSIGN_OR_RAISE(auto fn, reg->GetFunction("greater"));
    auto* sfn = static_cast<cp::ScalarFunction*>(fn.get());
    ARROW_ASSIGN_OR_RAISE(const cp::Kernel* k0, sfn->DispatchExact({arrow::float64(), arrow::float64()}));
    cp::ScalarKernel copy = *static_cast<const cp::ScalarKernel*>(k0);
    g_stock_greater.exec = copy.exec;
    g_stock_greater.init = copy.init;
    copy.exec = GreaterExecNP;
    copy.null_handling = cp::NullHandling::COMPUTED_NO_PREALLOCATE;
    copy.mem_allocation = cp::MemAllocation::NO_PREALLOCATE;
    ARROW_RETURN_NOT_OK(sfn->AddKernel(std::move(copy)));
  }
  ARROW_RETURN_NOT_OK(RegisterScalarBinaryNP<OpGreaterI64>(reg, "greater", arrow::int64()));
  ARROW_RETURN_NOT_OK(RegisterScalarBinaryNP<OpAddI64>(reg, "add", arrow::int64()));
  ARROW_RETURN_NOT_OK(RegisterScalarBinaryNP<OpAddF64>(reg, "add", arrow::float64()));
#define ARX_REGISTER_COMPARE(NAME, CMP, SLOT)                                                                          \
  ARROW_RETURN_NOT_OK((RegisterScalarBinaryNP<OpCompare<int64_t, arrow::Int64Scalar, CMP, SLOT>>(reg, NAME,          \
                                                                                               arrow::int64())));    \
  ARROW_RETURN_NOT_OK((RegisterScalarBinaryNP<OpCompare<double, arrow::DoubleScalar, CMP, SLOT + 1>>(reg, NAME,      \
                                                                                                  arrow::float64())))
  ARX_REGISTER_COMPARE("equal", ARX_CMP_EQUAL, 0);
  ARX_REGISTER_COMPARE("not_equal", ARX_CMP_NOT_EQUAL, 2);
  ARX_REGISTER_COMPARE("greater_equal", ARX_CMP_GREATER_EQUAL, 4);
  ARX_REGISTER_COMPARE("less", ARX_CMP_LESS, 6);
  ARX_REGISTER_COMPARE("less_equal", ARX_CMP_LESS_EQUAL, 8);
#undef ARX_REGISTER_COMPARE
#define ARX_REGISTER_ARITH(NAME, OP, CHECKED, SLOT)                                                                     \
  ARROW_RETURN_NOT_OK((RegisterScalarBinaryNP<OpArith<int64_t, arrow::Int64Scalar, OP, CHECKED, SLOT>>(reg, NAME,     \
                                                                                                     arrow::int64()))); \
  ARROW_RETURN_NOT_OK((RegisterScalarBinaryNP<OpArith<double, arrow::DoubleScalar, OP, CHECKED, SLOT + 1>>(            \
      reg, NAME, arrow::float64())))
  ARX_REGISTER_ARITH("subtract", ARX_ARITH_SUBTRACT, false, 0);
  ARX_REGISTER_ARITH("multiply", ARX_ARITH_MULTIPLY, false, 2);
  ARX_REGISTER_ARITH("add_checked", ARX_ARITH_ADD, true, 4);
  ARX_REGISTER_ARITH("subtract_checked", ARX_ARITH_SUBTRACT, true, 6);
  ARX_REGISTER_ARITH("multiply_checked", ARX_ARITH_MULTIPLY, true, 8);
#undef ARX_REGISTER_ARITH
  ARROW_RETURN_NOT_OK(RegisterBooleanNP(reg, "and_kleene", 2, KleeneExecNP<ARX_AND_KLEENE>, &g_stock_and_kleene));
  ARROW_RETURN_NOT_OK(RegisterBooleanNP(reg, "or_kleene", 2, KleeneExecNP<ARX_OR_KLEENE>, &g_stock_or_kleene));
  ARROW_RETURN_NOT_OK(RegisterBooleanNP(reg, "invert", 1, InvertExecNP, &g_stock_invert));
  {
    ARROW_ASSIGN_OR_RAISE(auto stock_cast, reg->GetFunction("cast"));
    ARROW_RETURN_NOT_OK(reg->AddFunction(std::make_shared<RocmCastMetaFunction>(std::move(stock_cast)),
                                         /*allow_overwrite=*/true));
  }
  {
    ARROW_ASSIGN_OR_RAISE(auto fn, reg->GetFunction("hash_sum"));
    if (fn->kind() != cp::Function::HASH_AGGREGATE) return Status::Invalid("hash_sum is not a hash aggregate");
    auto* hfn = static_cast<cp::HashAggregateFunction*>(fn.get());
    ARROW_ASSIGN_OR_RAISE(const cp::Kernel* k0, hfn->DispatchExact({arrow::int64(), arrow::uint32()}));
    cp::HashAggregateKernel copy = *static_cast<const cp::HashAggregateKernel*>(k0);
    // the stock signature's output resolver casts the kernel state to the reference's
    // GroupedAggregator (hash_aggregate_internal.h:88-91) — ours is not one: state the type
    // (FindAccumulatorType<Int64Type> = int64, aggregate_internal.h:41-44)
    copy.signature = cp::KernelSignature::Make({cp::InputType(arrow::int64()), cp::InputType(arrow::uint32())},
                                               cp::OutputType(arrow::int64()));
    copy.init = HashSumInit;
    copy.resize = HashSumResize;
    copy.consume = HashSumConsume;
    copy.merge = HashSumMerge;
    copy.finalize = HashSumFinalize;
    ARROW_RETURN_NOT_OK(hfn->AddKernel(std::move(copy)));
  }
  {
    const Status st = ac::default_exec_factory_registry()->AddFactory("aggregate_rocm", RocmGroupBySumNode::Make);
    if (!st.ok() && !st.IsKeyError()) return st;
  }
  return Status::OK();
}

std::once_flag g_once;
Status g_register_status;

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
// and binary / utf8 arrays without children.  0 on success.
int arrow_amd_copy_to_device(struct ArrowArray* in, struct ArrowSchema* schema, struct ArrowDeviceArray* out) {
  auto run = [&]() -> Status {
    ARROW_ASSIGN_OR_RAISE(auto host, arrow::ImportArray(in, schema));
    if (!host->data()->child_data.empty() || host->data()->dictionary != nullptr || host->data()->buffers.size() > 3) {
      return Status::NotImplemented("arrow_amd_copy_to_device: ", host->type()->ToString());
    }
    ARROW_ASSIGN_OR_RAISE(auto mm, RocmMemoryManagerFor(0));
    const int nbuf = static_cast<int>(host->data()->buffers.size());
    std::vector<std::shared_ptr<Buffer>> bufs(nbuf);
    for (int i = 0; i < nbuf; ++i) {
      const auto& b = host->data()->buffers[i];
      if (b != nullptr) {
        ARROW_ASSIGN_OR_RAISE(bufs[i], arrow::MemoryManager::CopyBuffer(b, mm));
      }
    }
    // null_count must be exact: nothing may popcount a device bitmap on the CPU later
    auto data = ArrayData::Make(host->type(), host->length(), std::move(bufs), host->null_count(), host->offset());
    return arrow::ExportDeviceArray(*arrow::MakeArray(data), nullptr, out);
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
    std::vector<std::shared_ptr<Buffer>> bufs(dev->data()->buffers.size());
    for (size_t i = 0; i < bufs.size(); ++i) {
      const auto& b = dev->data()->buffers[i];
      if (b != nullptr) {
        ARROW_ASSIGN_OR_RAISE(bufs[i], arrow::MemoryManager::CopyBuffer(b, arrow::default_cpu_memory_manager()));
      }
    }
    auto data = ArrayData::Make(dev->type(), dev->length(), std::move(bufs), dev->data()->null_count.load(),
                                dev->offset());
    return arrow::ExportArray(*arrow::MakeArray(data), out, out_schema);
  };
  const Status st = run();
  if (!st.ok()) {
    t_error = st.ToString();
    return -1;
  }
  return 0;
}
// The caching device allocator: bytes parked on free lists, allocations served from them / by hipMalloc.
void arrow_amd_plugin_pool_stats(int64_t* cached_bytes, int64_t* hits, int64_t* misses) {
  DevicePool::Get().Stats(cached_bytes, hits, misses);
}
// Returns every cached block to the driver.
void arrow_amd_plugin_pool_trim(void) { DevicePool::Get().Trim(); }
// Inputs shorter than this stay on the stock CPU kernels (PCIe staging does not pay).
void arrow_amd_plugin_set_min_rows(int64_t n) { g_min_rows.store(n); }
// The same threshold for the element-wise kernels (greater, cast); default: never stage them.
void arrow_amd_plugin_set_min_rows_streaming(int64_t n) { g_min_rows_streaming.store(n); }

}  // extern "C"
