// libsvm text -> CSR in HBM.  The reference re-opens and re-parses its part file on every ADMM
// iteration (src/admm_allreduce.cpp:57, src/workers.cc:74); here a part is parsed once and stays
// resident on the device.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <thread>

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

// Decimal text -> float.  Short plain decimals (at most 7 significant digits, at most 10 of them
// after the point, no exponent) are exact in float arithmetic -- an integer below 2^24 divided by
// a power of ten that float represents exactly is rounded once, as strtof rounds; everything
// else goes through strtof.
inline float parse_float(const char *b, const char *e) {
  static const float kPow10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
  const char *p = b;
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
  uint32_t mant = 0;
  int digits = 0, frac = 0;
  bool point = false, simple = p < e;
  for (; p < e; ++p) {
    const char c = *p;
    if (c >= '0' && c <= '9') {
      if (digits || c != '0') {
        if (++digits > 7) {
          simple = false;
          break;
        }
      }
      mant = mant * 10 + (uint32_t)(c - '0');
      if (point) ++frac;
    } else if (c == '.' && !point) {
      point = true;
    } else {
      simple = false;
      break;
    }
  }
  if (simple && frac <= 10) {
    const float v = (float)mant / kPow10[frac];
    return neg ? -v : v;
  }
  char buf[64];
  const size_t n = std::min<size_t>(e - b, sizeof(buf) - 1);
  std::memcpy(buf, b, n);
  buf[n] = 0;
  return std::strtof(buf, nullptr);
}

// what one thread parses: a run of whole lines
struct Chunk {
  std::vector<float> label, val;
  std::vector<uint32_t> col;
  std::vector<uint64_t> rowlen;
  uint32_t max_index = 0;
};

// the grammar: `label idx[:val] idx[:val] ...`, one row per non-blank line; an index alone has
// value 1
void parse_lines(const char *text, size_t b, size_t e, Chunk &out) {
  size_t p = b;
  while (p < e) {
    size_t q = p;
    while (q < e && text[q] != '\n') ++q;
    size_t s = p;
    while (s < q && is_blank(text[s])) ++s;
    if (s < q) {
      size_t t = s;
      while (t < q && !is_blank(text[t])) ++t;
      out.label.push_back(parse_float(text + s, text + t));
      const size_t before = out.col.size();
      s = t;
      while (s < q) {
        while (s < q && is_blank(text[s])) ++s;
        if (s >= q) break;
        t = s;
        uint64_t idx = 0;
        while (t < q && text[t] >= '0' && text[t] <= '9') idx = idx * 10 + (uint64_t)(text[t++] - '0');
        float v = 1.0f;
        if (t < q && text[t] == ':') {
          const size_t u = ++t;
          while (t < q && !is_blank(text[t])) ++t;
          v = parse_float(text + u, text + t);
        } else {
          while (t < q && !is_blank(text[t])) ++t;
        }
        ADMM_CHECK(idx < 0xFFFFFFFEull);
        out.col.push_back((uint32_t)idx);
        out.val.push_back(v);
        out.max_index = std::max(out.max_index, (uint32_t)idx);
        s = t;
      }
      out.rowlen.push_back(out.col.size() - before);
    }
    p = q + 1;
  }
}

int parse_threads() {
  if (const char *v = std::getenv("ADMM_PARSE_THREADS"))
    if (std::atoi(v) > 0) return std::atoi(v);
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(hw ? hw : 1, 16);
}

size_t parse_min_bytes() {  // text per parsing thread below which no further thread is started
  if (const char *v = std::getenv("ADMM_PARSE_MIN_BYTES"))
    if (std::atoll(v) > 0) return (size_t)std::atoll(v);
  return (size_t)1 << 20;
}

// ---- binary cache of a parsed part -----------------------------------------------------------
//
// With ADMM_CSR_CACHE_DIR set, a parsed part is written there as raw arrays behind a header that
// names the text it came from (size, modification time, part k of n); a later run that finds a
// matching file reads it instead of parsing.

struct CacheHeader {
  char magic[8];
  uint64_t text_bytes, text_mtime_ns, part, nparts, rows, nnz, max_index;
};

std::string cache_path(const std::string &uri, uint32_t part, uint32_t nparts) {
  const char *dir = std::getenv("ADMM_CSR_CACHE_DIR");
  if (!dir || !*dir) return "";
  uint64_t h = 1469598103934665603ull;  // FNV-1a of the path
  for (unsigned char c : uri) h = (h ^ c) * 1099511628211ull;
  char name[96];
  std::snprintf(name, sizeof(name), "/admmcsr_%016llx_%uof%u.bin", (unsigned long long)h, part, nparts);
  return std::string(dir) + name;
}

template <class T>
bool read_vec(std::FILE *f, std::vector<T> &v, uint64_t n) {
  v.resize(n);
  return n == 0 || std::fread(v.data(), sizeof(T), n, f) == n;
}

}  // namespace

bool SampleSet::ParseText(const std::string &uri, uint32_t part_index, uint32_t num_parts,
                          std::vector<uint64_t> &rowptr_, std::vector<uint32_t> &col_, std::vector<float> &val_,
                          std::vector<float> &label_, uint32_t *max_index_out) {
  rowptr_.assign(1, 0);
  col_.clear();
  val_.clear();
  label_.clear();
  int fd = ::open(uri.c_str(), O_RDONLY);
  if (fd < 0) {
    Runtime::Get().TrackerPrintf("Fail open %s\n", uri.c_str());
    return false;
  }
  struct stat st;
  ::fstat(fd, &st);
  const size_t len = (size_t)st.st_size;
  if (num_parts == 0) num_parts = 1;
  uint32_t max_index = 0;
  const uint64_t mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
  const std::string cpath = cache_path(uri, part_index, num_parts);
  bool from_cache = false;
  if (!cpath.empty()) {
    if (std::FILE *f = std::fopen(cpath.c_str(), "rb")) {
      CacheHeader h;
      if (std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "ADMMCSR1", 8) == 0 && h.text_bytes == len &&
          h.text_mtime_ns == mtime_ns && h.part == part_index && h.nparts == num_parts) {
        from_cache = read_vec(f, rowptr_, h.rows + 1) && read_vec(f, col_, h.nnz) && read_vec(f, val_, h.nnz) &&
                     read_vec(f, label_, h.rows);
        max_index = (uint32_t)h.max_index;
      }
      std::fclose(f);
      if (!from_cache) rowptr_.assign(1, 0), col_.clear(), val_.clear(), label_.clear();
    }
  }
  if (!from_cache) {
    const char *text = len ? (const char *)::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    if (len && text == MAP_FAILED) {
      ::close(fd);
      Runtime::Get().TrackerPrintf("Fail open %s\n", uri.c_str());
      return false;
    }
    // part k of n: the rows whose first byte lies in [k*step, (k+1)*step), step = ceil(len/n)
    const size_t step = (len + num_parts - 1) / num_parts;
    size_t b = std::min(len, (size_t)part_index * step), e = std::min(len, (size_t)(part_index + 1) * step);
    while (b > 0 && b < len && text[b - 1] != '\n') ++b;
    while (e > 0 && e < len && text[e - 1] != '\n') ++e;
    // the part is cut again, at line starts, into one run of lines per thread
    const int nthr = (int)std::max<size_t>(1, std::min<size_t>((size_t)parse_threads(), (e - b) / parse_min_bytes() + 1));
    std::vector<size_t> cut(nthr + 1);
    for (int t = 0; t <= nthr; ++t) {
      size_t c = b + (e - b) / nthr * t;
      if (t == nthr) c = e;
      while (c > b && c < e && text[c - 1] != '\n') ++c;
      cut[t] = c;
    }
    std::vector<Chunk> chunks(nthr);
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; ++t) pool.emplace_back(parse_lines, text, cut[t], cut[t + 1], std::ref(chunks[t]));
    parse_lines(text, cut[0], cut[1], chunks[0]);
    for (auto &th : pool) th.join();
    if (len) ::munmap((void *)text, len);
    // stitch the runs together
    uint64_t rows = 0, nnz = 0;
    std::vector<uint64_t> row0(nthr), nnz0(nthr);
    for (int t = 0; t < nthr; ++t) {
      row0[t] = rows;
      nnz0[t] = nnz;
      rows += chunks[t].label.size();
      nnz += chunks[t].col.size();
      max_index = std::max(max_index, chunks[t].max_index);
    }
    rowptr_.resize(rows + 1);
    label_.resize(rows);
    col_.resize(nnz);
    val_.resize(nnz);
    auto stitch = [&](int t) {
      Chunk &c = chunks[t];
      if (!c.label.empty()) std::memcpy(label_.data() + row0[t], c.label.data(), sizeof(float) * c.label.size());
      if (!c.col.empty()) {
        std::memcpy(col_.data() + nnz0[t], c.col.data(), sizeof(uint32_t) * c.col.size());
        std::memcpy(val_.data() + nnz0[t], c.val.data(), sizeof(float) * c.val.size());
      }
      uint64_t at = nnz0[t];
      for (size_t r = 0; r < c.rowlen.size(); ++r) {
        rowptr_[row0[t] + r] = at;
        at += c.rowlen[r];
      }
      c = Chunk();  // the run's own arrays are not needed any more
    };
    pool.clear();
    for (int t = 1; t < nthr; ++t) pool.emplace_back(stitch, t);
    stitch(0);
    for (auto &th : pool) th.join();
    rowptr_[rows] = nnz;
    if (!cpath.empty()) {
      const std::string tmp = cpath + ".tmp" + std::to_string((long)::getpid());
      if (std::FILE *f = std::fopen(tmp.c_str(), "wb")) {
        CacheHeader h = {{'A', 'D', 'M', 'M', 'C', 'S', 'R', '1'}, len, mtime_ns, part_index, num_parts, rows, nnz, max_index};
        bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
        ok = ok && std::fwrite(rowptr_.data(), sizeof(uint64_t), rows + 1, f) == rows + 1;
        ok = ok && (nnz == 0 || std::fwrite(col_.data(), sizeof(uint32_t), nnz, f) == nnz);
        ok = ok && (nnz == 0 || std::fwrite(val_.data(), sizeof(float), nnz, f) == nnz);
        ok = ok && (rows == 0 || std::fwrite(label_.data(), sizeof(float), rows, f) == rows);
        ok = (std::fclose(f) == 0) && ok;
        if (ok) std::rename(tmp.c_str(), cpath.c_str());
        else std::remove(tmp.c_str());
      }
    }
  }
  ::close(fd);
  *max_index_out = max_index;
  return true;
}

bool SampleSet::Initialize(const std::string &uri, uint32_t part_index, uint32_t num_parts, size_t dim) {
  Release();
  rows_ = nnz_ = 0;
  rb_size_ = -1;
  current_index_ = -1;
  block_served_ = false;
  uint32_t max_index = 0;
  if (!ParseText(uri, part_index, num_parts, rowptr_, col_, val_, label_, &max_index)) return false;
  rows_ = label_.size();
  nnz_ = col_.size();
  if (dim == 0) dim = (size_t)max_index + 2;
  const int rc = admm_csr_upload(Runtime::Get().ctx(), rowptr_.data(), col_.data(), val_.data(), label_.data(), rows_,
                                 nnz_, dim, /*add_bias=*/1, &csr_);
  if (rc != ADMM_OK) {
    // an index beyond dim is the reference's CHECK failure inside SDot
    std::fprintf(stderr, "%s\n", admm_last_error());
    std::abort();
  }
  // The parsed arrays live in HBM from here on; the host copies (12 bytes per entry, 80 GB for a 10^10-entry
  // shard) are given back.  The row cursor fetches them from the device again if anybody uses it.
  std::vector<uint64_t>().swap(rowptr_);
  std::vector<uint32_t>().swap(col_);
  std::vector<float>().swap(val_);
  std::vector<float>().swap(label_);
  std::vector<size_t>().swap(col_wide_);
  return true;
}

void SampleSet::EnsureHost() {
  if (!rowptr_.empty() || csr_ == nullptr) return;
  uint64_t nr = 0, nz = 0, dm = 0;
  ADMM_CALL(admm_csr_shape(csr_, &nr, &nz, &dm));
  std::vector<uint64_t> rp(nr + 1);
  std::vector<uint32_t> c(nz);
  std::vector<float> v(nz);
  label_.resize(nr);
  ADMM_CALL(admm_csr_download(csr_, rp.data(), c.data(), v.data(), label_.data()));
  // the device holds translated rows (bias column first, indices shifted by one): undo that
  rowptr_.resize(nr + 1);
  col_.resize(nz - nr);
  val_.resize(nz - nr);
  for (uint64_t r = 0; r < nr; ++r) {
    rowptr_[r] = rp[r] - r;
    for (uint64_t k = rp[r] + 1; k < rp[r + 1]; ++k) {
      col_[k - r - 1] = c[k] - 1;
      val_[k - r - 1] = v[k];
    }
  }
  rowptr_[nr] = rp[nr] - nr;
}

void SampleSet::PrepareTransposed(uint64_t panel_rows, uint32_t col_block, bool relabel, bool lean) {
  ADMM_CHECK(csr_ != nullptr);
  // lean: the set will only be swept as a whole from here on (the L-BFGS local solver); the library then frees
  // each row-major form once the copy that replaces it exists, and the row cursor has nothing left to read
  if (lean) ADMM_CALL(admm_csr_set_lean(csr_, 1));
  // relabelling first: the copies are then built once, in the final numbering
  if (relabel) ADMM_CALL(admm_csr_relabel_by_frequency(csr_));
  ADMM_CALL(admm_csr_build_transposed(csr_, panel_rows, col_block));
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
  EnsureHost();
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
