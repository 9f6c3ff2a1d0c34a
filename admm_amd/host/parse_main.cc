// ./parsecheck <file> <part> <nparts> [-q]
//
// Prints what the loader makes of part `part` of `nparts` of a libsvm text file: a line
// "rows nnz max_index", then every row as "label idx:val ..." with nine significant digits.
// With -q only the first line and the parse time are printed.  No GPU is touched; the tests compare
// the dump with their own reading of the file.  With -c the part is loaded into HBM (SampleSet::Initialize)
// and the rows are printed through the row cursor (Next / GetData), which fetches them back from the device.
#include <chrono>
#include <cstring>

#include "admm_host.h"

int main(int argc, char *argv[]) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <file> <part> <nparts>\n", argv[0]);
    return 1;
  }
  if (argc > 4 && std::strcmp(argv[4], "-c") == 0) {
    admm::Runtime::Get().Init(argc, argv);
    {
      admm::SampleSet set;
      if (!set.Initialize(argv[1], (uint32_t)std::atoi(argv[2]), (uint32_t)std::atoi(argv[3]))) return 2;
      std::printf("%d %llu\n", set.Size(), (unsigned long long)set.nnz_with_bias());
      set.Rewind();
      while (set.Next()) {
        const admm::Row r = set.GetData();
        std::printf("%.9g", r.label);
        for (size_t k = 0; k < r.length; ++k) std::printf(" %zu:%.9g", r.index[k], r.value[k]);
        std::printf("\n");
      }
    }
    admm::Runtime::Get().Finalize();
    return 0;
  }
  std::vector<uint64_t> rowptr;
  std::vector<uint32_t> col;
  std::vector<float> val, label;
  uint32_t max_index = 0;
  const auto t0 = std::chrono::steady_clock::now();
  if (!admm::SampleSet::ParseText(argv[1], (uint32_t)std::atoi(argv[2]), (uint32_t)std::atoi(argv[3]), rowptr, col,
                                  val, label, &max_index))
    return 2;
  std::printf("%zu %zu %u\n", label.size(), col.size(), max_index);
  if (argc > 4 && std::strcmp(argv[4], "-q") == 0) {
    std::printf("%.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 0;
  }
  for (size_t r = 0; r < label.size(); ++r) {
    std::printf("%.9g", label[r]);
    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) std::printf(" %u:%.9g", col[k], val[k]);
    std::printf("\n");
  }
  return 0;
}
