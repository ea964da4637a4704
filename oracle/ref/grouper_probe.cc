// TEST INFRASTRUCTURE ONLY.  Runs the REFERENCE's own Grouper (arrow::compute::Grouper::Make -> GrouperFastImpl for
// fixed-width keys, cpp/src/arrow/compute/row/grouper.cc:555-973) from the installed pyarrow wheel's libarrow_compute on
// key columns read from a file, so that oracle.Grouper (the restatement the device Grouper is tested against) is pinned
// to the reference implementation itself and not only to the expectations transcribed from grouper_test.cc.
//
// input file:  int64 n, int64 ncols, int64 batch_rows, then per column: int64 byte_width, int64 is_float,
//              n * byte_width value bytes, n validity bytes (1 = valid)
// output file: int64 num_groups, n uint32 ids (Consume of every batch in order), then per column:
//              num_groups * byte_width unique value bytes, num_groups validity bytes
// Built by tests/test_oracle_pin.py with g++ against the wheel's headers; never shipped, never linked by the product.
#include <arrow/api.h>
#include <arrow/compute/api.h>
#include <arrow/compute/initialize.h>
#include <arrow/compute/row/grouper.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

using arrow::Status;

static std::shared_ptr<arrow::DataType> TypeOf(int64_t width, int64_t is_float) {
  if (is_float) return width == 4 ? arrow::float32() : arrow::float64();
  switch (width) {
    case 1: return arrow::uint8();
    case 2: return arrow::uint16();
    case 4: return arrow::uint32();
    default: return arrow::uint64();
  }
}

static Status Run(const char* in_path, const char* out_path) {
  ARROW_RETURN_NOT_OK(arrow::compute::Initialize());
  FILE* f = std::fopen(in_path, "rb");
  if (!f) return Status::IOError("cannot open ", in_path);
  int64_t hdr[3];
  if (std::fread(hdr, 8, 3, f) != 3) return Status::IOError("short header");
  const int64_t n = hdr[0], ncols = hdr[1], batch_rows = hdr[2];
  std::vector<std::shared_ptr<arrow::Array>> cols;
  std::vector<arrow::TypeHolder> types;
  std::vector<int64_t> widths;
  for (int64_t c = 0; c < ncols; ++c) {
    int64_t meta[2];
    if (std::fread(meta, 8, 2, f) != 2) return Status::IOError("short column header");
    const int64_t w = meta[0];
    ARROW_ASSIGN_OR_RAISE(auto data, arrow::AllocateBuffer(n * w + 8));
    std::vector<uint8_t> valid(n);
    if (n && (std::fread(data->mutable_data(), 1, n * w, f) != static_cast<size_t>(n * w) ||
              std::fread(valid.data(), 1, n, f) != static_cast<size_t>(n))) {
      return Status::IOError("short column");
    }
    ARROW_ASSIGN_OR_RAISE(auto bitmap, arrow::AllocateEmptyBitmap(n));
    int64_t nulls = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (valid[i]) arrow::bit_util::SetBit(bitmap->mutable_data(), i); else ++nulls;
    }
    auto type = TypeOf(w, meta[1]);
    cols.push_back(arrow::MakeArray(arrow::ArrayData::Make(type, n, {std::move(bitmap), std::move(data)}, nulls)));
    types.emplace_back(type);
    widths.push_back(w);
  }
  std::fclose(f);
  ARROW_ASSIGN_OR_RAISE(auto grouper, arrow::compute::Grouper::Make(types));
  std::vector<uint32_t> ids(n);
  for (int64_t b = 0; b < n; b += batch_rows) {
    const int64_t m = std::min(batch_rows, n - b);
    std::vector<arrow::Datum> values;
    for (auto& c : cols) values.emplace_back(c->Slice(b, m));
    arrow::compute::ExecBatch batch(std::move(values), m);
    ARROW_ASSIGN_OR_RAISE(arrow::Datum out, grouper->Consume(arrow::compute::ExecSpan(batch)));
    auto arr = out.make_array();
    std::memcpy(ids.data() + b, arr->data()->GetValues<uint32_t>(1), m * 4);
  }
  ARROW_ASSIGN_OR_RAISE(arrow::compute::ExecBatch uniques, grouper->GetUniques());
  const int64_t g = grouper->num_groups();
  FILE* o = std::fopen(out_path, "wb");
  if (!o) return Status::IOError("cannot open ", out_path);
  std::fwrite(&g, 8, 1, o);
  std::fwrite(ids.data(), 4, n, o);
  for (int64_t c = 0; c < ncols; ++c) {
    auto arr = uniques[c].make_array();
    const uint8_t* vals = arr->data()->buffers[1]->data() + arr->offset() * widths[c];
    std::fwrite(vals, 1, g * widths[c], o);
    std::vector<uint8_t> valid(g);
    for (int64_t i = 0; i < g; ++i) valid[i] = arr->IsValid(i) ? 1 : 0;
    std::fwrite(valid.data(), 1, g, o);
  }
  std::fclose(o);
  return Status::OK();
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  Status st = Run(argv[1], argv[2]);
  if (!st.ok()) {
    std::fprintf(stderr, "%s\n", st.ToString().c_str());
    return 1;
  }
  return 0;
}
