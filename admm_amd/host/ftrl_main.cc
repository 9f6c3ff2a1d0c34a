// ./ftrl <conf>
//
// Stand-alone FTRL-proximal training per rank (src/ftrl_main.cpp:106-156): `passes` passes over
// the rank's parts, logloss / AUC on its test file after each pass, then ftrl_weight_<psid>
// (three fixed-width columns w, z, n) and <psid>.bin (1-based `i:w`).  Paths are concatenated
// without a separator, as the reference does.
#include <fstream>
#include <iomanip>

#include "admm_host.h"

using admm::Runtime;

namespace {

bool load_state(const std::string &path, ::ftrl::FtrlSolver *s) {  // src/ftrl_main.cpp:29-53
  std::ifstream is(path);
  if (!is.is_open()) return false;
  std::vector<float> w, z, n;
  std::string a, b, c;
  while (is >> a >> b >> c) {
    w.push_back((float)std::atof(a.c_str()));
    z.push_back((float)std::atof(b.c_str()));
    n.push_back((float)std::atof(c.c_str()));
  }
  s->weight_.from_host(w);
  s->mid_weight_.from_host(z);
  s->squared_sum_.from_host(n);
  return true;
}

}  // namespace

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <conf>\n", argv[0]);
    return 1;
  }
  Runtime &rt = Runtime::Get();
  rt.Init(argc, argv);
  {
    Metrics metrics;
    ::admm::ArgParser arg_parser;
    ::admm::FtrlConfig prm;
    arg_parser.FTRLParse(argv[1], prm);
    ADMM_CHECK(prm.dim > 0);
    ADMM_CHECK((size_t)rt.GetRank() < arg_parser.psid.size());
    ::ftrl::FtrlSolver solver;
    solver.Init(prm);
    solver.psid_ = arg_parser.psid[rt.GetRank()];
    solver.num_part_ = arg_parser.num_part[rt.GetRank()];
    if (!arg_parser.load_path.empty()) load_state(prm.output_path + "ftrl_weight_" + solver.psid_, &solver);

    const std::string train_name = prm.train_path + solver.psid_ + "_aggregated/part-00000";
    const std::string test_name = prm.test_path + solver.psid_ + "_aggregated/part-00000";
    ::admm::DeviceVector offset(prm.dim), reg_offset(prm.dim);

    std::vector<std::unique_ptr<::admm::SampleSet>> parts;
    for (int part = 0; part < solver.num_part_; ++part) {
      parts.emplace_back(new ::admm::SampleSet);
      ADMM_CHECK(parts.back()->Initialize(train_name, part, solver.num_part_, prm.dim));
      if (prm.ftrl_batch > 1) parts.back()->PrepareTransposed(prm.ftrl_batch, 0, true);
    }
    ::admm::SampleSet test_set;
    ADMM_CHECK(test_set.Initialize(test_name, 0, 1, prm.dim));

    rt.TrackerPrintf("ftrl execution \n");
    for (int i = 0; i < prm.passes; ++i) {
      for (auto &part : parts) solver.Run(*part, test_set, offset, reg_offset);
      std::vector<const ::admm::DeviceVector *> group_w = {&solver.weight_};
      float loss = 0, auc = 0;
      metrics.Both(test_set, group_w, &loss, &auc);
      rt.TrackerPrintf("%s | %.7f | %.7f | %d\n", solver.psid_.c_str(), loss, auc, test_set.Size());
    }

    const std::vector<float> w = solver.weight_.to_host(), z = solver.mid_weight_.to_host(),
                             n = solver.squared_sum_.to_host();
    {
      std::ofstream os(prm.output_path + "ftrl_weight_" + solver.psid_);
      ADMM_CHECK(os.is_open());
      for (size_t i = 0; i < solver.dim_; ++i)
        os << std::left << std::setw(17) << w[i] << std::left << std::setw(17) << z[i] << std::left << std::setw(17)
           << n[i] << '\n';
    }
    {
      std::ofstream os(prm.output_path + solver.psid_ + ".bin");
      ADMM_CHECK(os.is_open());
      for (size_t i = 0; i < w.size(); ++i) os << i + 1 << ":" << w[i] << '\n';
    }
  }
  rt.Finalize();
  return 0;
}
