// ./admm <conf> [<train_dir> <test_dir> <out_dir> [load_dir]]
//
// The ADMM outer loop of the reference (src/admm_allreduce.cpp:7-115) with every vector in HBM:
// local x-update, payload rho*x + y, sum all-reduce, test metrics, consensus z-update, dual
// update, and the three state files at the end.  With only <conf> given, the directories come
// from the conf keys train_path / test_path / output_path (and load_path).
#include <cmath>

#include "admm_host.h"

using admm::Runtime;

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <conf> [<train_dir> <test_dir> <out_dir> [load_dir]]\n", argv[0]);
    return 1;
  }
  Runtime &rt = Runtime::Get();
  rt.Init(argc, argv);
  {
    LocalModel local_model;
    GlobalModel global_model;
    Metrics metrics;
    ::admm::Master master_processor;
    ::admm::ArgParser arg_parser;

    global_model.InitModel(argv[1], arg_parser);
    ::admm::AdmmConfig &prm = global_model.admm_params_;
    if (argc >= 5) {
      prm.train_path = argv[2];
      prm.test_path = argv[3];
      prm.output_path = argv[4];
    }
    ADMM_CHECK(prm.dim > 0);
    ADMM_CHECK(!prm.train_path.empty() && !prm.output_path.empty());
    local_model.InitModel(prm.dim, arg_parser);
    ::admm::Worker &worker = local_model.worker_processor_;

    const std::string load_dir = argc > 5 ? argv[5] : arg_parser.load_path;
    if (!load_dir.empty()) {  // warm start, src/admm_allreduce.cpp:25-45
      const std::string local_load = load_dir + "/admm_weight_" + worker.psid_;
      if (!local_model.Load(local_load)) rt.TrackerPrintf("Fail load file %s\n", local_load.c_str());
      const std::string global_load = load_dir + "/global_params_" + worker.psid_;
      if (!global_model.Load(global_load)) rt.TrackerPrintf("Fail load file %s\n", global_load.c_str());
    }

    const std::string test_name = prm.test_path + "/" + worker.psid_ + "_aggregated/part-00000";
    ::admm::SampleSet train_unused, test_set;
    const bool have_test = test_set.Initialize(test_name, 0, 1, prm.dim);

    const int max_iter = prm.passes;
    const std::size_t dim = prm.dim;
    // The payload buffer carries four more floats: this rank's residual sums |x - z|^2 and
    // rho^2 |z - z_prev|^2 of the PREVIOUS iteration (they need the z the all-reduce produces), each as a
    // (high, low) pair, so that the iteration's one all-reduce also yields the residuals over all ranks.
    constexpr std::size_t kTail = 4;
    ::admm::DeviceVector tmp(dim + kTail);
    double pend_r2 = 0, pend_s2 = 0;
    bool pending = false;
    auto tail_up = [&]() {
      float t[kTail] = {0, 0, 0, 0};
      if (pending) {
        t[0] = (float)pend_r2;
        t[1] = (float)(pend_r2 - (double)t[0]);
        t[2] = (float)pend_s2;
        t[3] = (float)(pend_s2 - (double)t[2]);
      }
      ADMM_CALL(admm_vec_upload(rt.ctx(), tmp.data() + dim, t, kTail));
    };
    auto tail_down = [&](int iter) {
      if (!pending) return;
      float t[kTail];
      ADMM_CALL(admm_vec_download(rt.ctx(), t, tmp.data() + dim, kTail));
      const double r2 = (double)t[0] + (double)t[1];
      const double s2 = ((double)t[2] + (double)t[3]) / rt.GetWorldSize();  // every rank holds the same s2
      if (rt.GetRank() == 0)
        rt.TrackerPrintf("[residual-all] %d %.7e %.7e\n", iter, std::sqrt(r2), std::sqrt(s2));
      pending = false;
    };

    for (int r = 0; r < max_iter; ++r) {
      if (r == max_iter - 1) worker.load = true;
      // the prepare step of rabit::Allreduce (lazy_ftrl), then the reduction itself
      worker.BaseUpdate(train_unused, test_set, prm);
      worker.GetWeights(prm, tmp);
      tail_up();
      rt.AllreduceSum(tmp.data(), dim + kTail);
      tail_down(r - 1);

      if (have_test) {
        std::vector<const ::admm::DeviceVector *> group_w = {&worker.base_vec_, &worker.bias_vec_};
        float loss = 0, auc = 0;
        metrics.Both(test_set, group_w, &loss, &auc);
        rt.TrackerPrintf("%s | %.7f | %.7f | %d\n", worker.psid_.c_str(), loss, auc, test_set.Size());
      }
      if (prm.global_metrics) {  // conf key of this build: the same two numbers over all ranks' test rows
        std::vector<const ::admm::DeviceVector *> group_w = {&worker.base_vec_, &worker.bias_vec_};
        double gl = 0, ga = 0;
        uint64_t gn = 0;
        metrics.AllRanks(rt, test_set, group_w, &gl, &ga, &gn);
        if (rt.GetRank() == 0) rt.TrackerPrintf("all | %.7f | %.7f | %llu\n", gl, ga, (unsigned long long)gn);
      }

      double r2 = 0, s2 = 0;  // GlobalUpdate + LangrangeUpdate, fused
      ADMM_CHECK(master_processor.Consensus(tmp, prm, rt.GetWorldSize(), worker, &r2, &s2));
      rt.TrackerPrintf("[residual] %s %d %.7e %.7e\n", worker.psid_.c_str(), r, std::sqrt(r2), std::sqrt(s2));
      pend_r2 = r2;
      pend_s2 = s2;
      pending = true;
      if (rt.GetRank() == 0) rt.TrackerPrintf("Finish %d-th iteration\n", r);
    }
    if (pending) {  // the last iteration's residuals over all ranks: one more exchange of the tail alone
      tail_up();
      rt.AllreduceSum(tmp.data() + dim, kTail);
      tail_down(max_iter - 1);
    }

    local_model.SaveState(prm.output_path + "/admm_weight_" + worker.psid_);
    local_model.Save(prm.output_path + "/" + worker.psid_ + ".bin");
    global_model.Save(prm.output_path + "/global_params_" + worker.psid_);
  }
  rt.Finalize();
  return 0;
}
