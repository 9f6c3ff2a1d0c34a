// ./evan <weights> <train_dir> <test_dir> <id>...
//
// Logloss / AUC of a dense whitespace-separated weight file on rank r's train and test files
// (src/eval_main.cpp:36-69).  The reference computes the four numbers and drops them; this
// program prints them.
#include <fstream>

#include "admm_host.h"

using admm::Runtime;

int main(int argc, char *argv[]) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s <weights> <train_dir> <test_dir> <id>...\n", argv[0]);
    return 1;
  }
  Runtime &rt = Runtime::Get();
  rt.Init(argc, argv);
  {
    rt.TrackerPrintf("Initialization \n");
    ADMM_CHECK(rt.GetRank() + 4 < argc);
    const std::string id = argv[rt.GetRank() + 4];
    const std::string train_name = std::string(argv[2]) + id, test_name = std::string(argv[3]) + id;
    std::vector<float> weights;
    {
      std::ifstream is(argv[1]);
      ADMM_CHECK(is.is_open());
      float v;
      while (is >> v) weights.push_back(v);
    }
    rt.TrackerPrintf("%d dim\n", (int)weights.size());
    ADMM_CHECK(!weights.empty());
    ::admm::SampleSet train_set, test_set;
    ADMM_CHECK(train_set.Initialize(train_name, 0, 1, weights.size()));
    ADMM_CHECK(test_set.Initialize(test_name, 0, 1, weights.size()));
    ::admm::DeviceVector w;
    w.from_host(weights);
    Metrics metrics;
    std::vector<const ::admm::DeviceVector *> group_w = {&w};
    rt.TrackerPrintf("%d task -- %s metrics: \n", rt.GetRank(), id.c_str());
    float loss = 0, auc = 0;
    metrics.Both(train_set, group_w, &loss, &auc);
    rt.TrackerPrintf("train | %.7f | %.7f | %d\n", loss, auc, train_set.Size());
    metrics.Both(test_set, group_w, &loss, &auc);
    rt.TrackerPrintf("test | %.7f | %.7f | %d\n", loss, auc, test_set.Size());
  }
  rt.Finalize();
  return 0;
}
