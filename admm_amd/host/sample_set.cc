// libsvm text -> CSR in HBM.  The reference re-opens and re-parses its part file on every ADMM
// iteration (src/admm_allreduce.cpp:57, src/workers.cc:74); here a part is parsed once and stays
// resident on the device.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

#include "admm_host.h"

namespace admm {

SampleSet::SampleSet() {}
SampleSet::~SampleSet() { Release(); }

void SampleSet::Release() {
  if (csr_) admm_csr_destroy(csr_);
  csr_ = nullptr;
}

namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// minimal float parser for the common libsvm shapes (digits, sign, '.', exponent); anything
// else goes through strtof
inline float parse_float(const char *b, const char *e) {
  char buf[64];
  size_t n = std::min<size_t>(e - b, sizeof(buf) - 1);
  std::memcpy(buf, b, n);
  buf[n] = 0;
  return std::strtof(buf, nullptr);
}

}  // namespace

bool SampleSet::Initialize(const std::string &uri, uint32_t part_index, uint32_t num_parts, size_t dim) {
  Release();
  rowptr_.assign(1, 0);
  col_.clear();
  val_.clear();
  label_.clear();
  rows_ = nnz_ = 0;
  rb_size_ = -1;
  current_index_ = -1;
  block_served_ = false;
  int fd = ::open(uri.c_str(), O_RDONLY);
  if (fd < 0) {
    Runtime::Get().TrackerPrintf("Fail open %s\n", uri.c_str());
    return false;
  }
  struct stat st;
  ::fstat(fd, &st);
  const size_t len = (size_t)st.st_size;
  const char *text = len ? (const char *)::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : "";
  if (len && text == MAP_FAILED) {
    ::close(fd);
    Runtime::Get().TrackerPrintf("Fail open %s\n", uri.c_str());
    return false;
  }
  // part k of n: the rows whose first byte lies in [k*step, (k+1)*step), step = ceil(len/n)
  if (num_parts == 0) num_parts = 1;
  const size_t step = (len + num_parts - 1) / num_parts;
  size_t b = std::min(len, (size_t)part_index * step), e = std::min(len, (size_t)(part_index + 1) * step);
  while (b > 0 && b < len && text[b - 1] != '\n') ++b;
  while (e > 0 && e < len && text[e - 1] != '\n') ++e;
  bool all_ones = true;
  uint32_t max_index = 0;
  size_t p = b;
  while (p < e) {
    size_t q = p;
    while (q < e && text[q] != '\n') ++q;
    size_t s = p;
    while (s < q && is_blank(text[s])) ++s;
    if (s < q) {
      size_t t = s;
      while (t < q && !is_blank(text[t])) ++t;
      label_.push_back(parse_float(text + s, text + t));
      s = t;
      while (s < q) {
        while (s < q && is_blank(text[s])) ++s;
        if (s >= q) break;
        t = s;
        uint64_t idx = 0;
        while (t < q && text[t] >= '0' && text[t] <= '9') idx = idx * 10 + (uint64_t)(text[t++] - '0');
        float v = 1.0f;
        if (t < q && text[t] == ':') {
          size_t u = ++t;
          while (t < q && !is_blank(text[t])) ++t;
          v = parse_float(text + u, text + t);
        } else {
          while (t < q && !is_blank(text[t])) ++t;
        }
        ADMM_CHECK(idx < 0xFFFFFFFEull);
        col_.push_back((uint32_t)idx);
        val_.push_back(v);
        all_ones = all_ones && v == 1.0f;
        max_index = std::max(max_index, (uint32_t)idx);
        s = t;
      }
      rowptr_.push_back(col_.size());
    }
    p = q + 1;
  }
  if (len) ::munmap((void *)text, len);
  ::close(fd);
  rows_ = label_.size();
  nnz_ = col_.size();
  if (dim == 0) dim = (size_t)max_index + 2;
  (void)all_ones;
  const int rc = admm_csr_upload(Runtime::Get().ctx(), rowptr_.data(), col_.data(), val_.data(), label_.data(), rows_,
                                 nnz_, dim, /*add_bias=*/1, &csr_);
  if (rc != ADMM_OK) {
    // an index beyond dim is the reference's CHECK failure inside SDot
    std::fprintf(stderr, "%s\n", admm_last_error());
    std::abort();
  }
  return true;
}

void SampleSet::PrepareTransposed(uint64_t panel_rows, uint32_t col_block, bool relabel) {
  ADMM_CHECK(csr_ != nullptr);
  ADMM_CALL(admm_csr_build_transposed(csr_, panel_rows, col_block));
  if (relabel) ADMM_CALL(admm_csr_relabel_by_frequency(csr_));
}

// Row cursor over the single in-memory block (src/sample_set.cc:32-68)
bool SampleSet::Next() {
  ++current_index_;
  if (current_index_ >= rb_size_) {
    if (block_served_) return false;
    block_served_ = true;
    rb_size_ = (int)rows_;
    current_index_ = 0;
    if (rows_ == 0) return false;
  }
  return true;
}

Row SampleSet::GetData() {
  Row r;
  const uint64_t b = rowptr_[current_index_], e = rowptr_[current_index_ + 1];
  if (col_wide_.size() != col_.size()) col_wide_.assign(col_.begin(), col_.end());
  r.label = label_[current_index_];
  r.length = e - b;
  r.index = col_wide_.data() + b;
  r.value = val_.data() + b;
  return r;
}

void SampleSet::Rewind() {
  current_index_ = -1;
  rb_size_ = -1;
  block_served_ = false;
}

Row SampleSet::GetLastData() {
  Row last;
  if (current_index_ >= 0 && current_index_ == rb_size_) {
    --current_index_;
    last = GetData();
    ++current_index_;
  } else {
    std::fprintf(stderr, "failed to get the last data. \n");
  }
  return last;
}

Row SampleSet::TranslateData(const Row &x) {
  Row t;
  t.label = x.label;
  t.weight = x.weight;
  t.length = 1 + x.length;
  size_t *idx = new size_t[t.length];
  float *val = new float[t.length];
  idx[0] = 0;
  val[0] = 1;
  for (size_t i = 0; i < x.length; ++i) {
    idx[i + 1] = x.index[i] + 1;
    val[i + 1] = x.value[i];
  }
  t.index = idx;
  t.value = val;
  return t;
}

// Rows of the current block.  The device-side metrics do not walk the cursor, so a set that has
// been loaded but not iterated reports its row count instead of the reference's -1.
int SampleSet::Size() { return rb_size_ >= 0 ? rb_size_ : (csr_ ? (int)rows_ : -1); }

}  // namespace admm
