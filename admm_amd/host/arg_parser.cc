// The conf-file grammar of src/arg_parser.cc:10-90: whitespace-separated tokens, `key = value`
// pairs go to a dictionary, `psid : num_part` lines list the workers (rank r uses line r).
#include <fstream>
#include <map>

#include "admm_host.h"

namespace admm {

namespace {

bool read_conf(const char *file, std::map<std::string, std::string> *dict, std::vector<std::string> *psid,
               std::vector<int> *num_part) {
  std::ifstream in(file);
  if (!in.is_open()) {
    Runtime::Get().TrackerPrintf("config file open failed. \n");
    return false;
  }
  std::string key, op, value;
  while (in >> key >> op) {
    if (op == "=") {
      if (!(in >> value)) break;
      (*dict)[key] = value;
    } else if (op == ":") {
      if (!(in >> value)) break;
      psid->push_back(key);
      num_part->push_back(std::atoi(value.c_str()));
    }
    // any other pair of tokens is skipped, as the reference's loop does
  }
  return true;
}

const char *get(const std::map<std::string, std::string> &d, const char *k) {
  auto it = d.find(k);
  return it == d.end() ? "" : it->second.c_str();
}

}  // namespace

void ArgParser::ADMMParse(const char *file, AdmmConfig &p) {
  std::map<std::string, std::string> d;
  if (!read_conf(file, &d, &psid, &num_part)) return;
  p.step_size = (float)std::atof(get(d, "step_size"));
  p.bias_var = (float)std::atof(get(d, "bias_var"));
  p.global_var = (float)std::atof(get(d, "global_var"));
  p.ftrl_alpha = (float)std::atof(get(d, "ftrl_alpha"));
  p.passes = std::atoi(get(d, "passes"));
  p.dim = static_cast<size_t>(std::atoi(get(d, "dim")));
  p.global_weights.resize(p.dim);
  // optional keys (the reference keeps the three path keys commented out and takes argv instead)
  if (d.count("train_path")) p.train_path = d["train_path"];
  if (d.count("test_path")) p.test_path = d["test_path"];
  if (d.count("output_path")) p.output_path = d["output_path"];
  if (d.count("global_l1")) p.global_l1 = (float)std::atof(d["global_l1"].c_str());
  if (d.count("global_metrics")) p.global_metrics = std::atoi(d["global_metrics"].c_str());
  if (d.count("local_solver")) p.local_solver = d["local_solver"];
  if (d.count("ftrl_batch")) p.ftrl_batch = std::max(1, std::atoi(d["ftrl_batch"].c_str()));
  if (d.count("lbfgs_m")) p.lbfgs.m = std::atoi(d["lbfgs_m"].c_str());
  if (d.count("lbfgs_max_iter")) p.lbfgs.max_iter = std::atoi(d["lbfgs_max_iter"].c_str());
  if (d.count("lbfgs_max_linesearch")) p.lbfgs.max_linesearch = std::atoi(d["lbfgs_max_linesearch"].c_str());
  if (d.count("lbfgs_epsilon")) p.lbfgs.epsilon = (float)std::atof(d["lbfgs_epsilon"].c_str());
  if (d.count("lbfgs_ftol")) p.lbfgs.ftol = (float)std::atof(d["lbfgs_ftol"].c_str());
  if (d.count("lbfgs_wolfe")) p.lbfgs.wolfe = (float)std::atof(d["lbfgs_wolfe"].c_str());
  if (d.count("lbfgs_linesearch")) p.lbfgs.linesearch = std::atoi(d["lbfgs_linesearch"].c_str());
  if (d.count("load_path")) load_path = d["load_path"];
}

void ArgParser::FTRLParse(const char *file, FtrlConfig &p) {
  std::map<std::string, std::string> d;
  if (!read_conf(file, &d, &psid, &num_part)) return;
  p.train_path = get(d, "train_path");
  p.test_path = get(d, "test_path");
  p.output_path = get(d, "output_path");
  p.alpha = (float)std::atof(get(d, "ftrl_alpha"));
  p.beta = (float)std::atof(get(d, "ftrl_beta"));
  p.l_1 = (float)std::atof(get(d, "l_1"));
  p.l_2 = (float)std::atof(get(d, "l_2"));
  p.passes = std::atoi(get(d, "passes"));
  p.dim = static_cast<size_t>(std::atoi(get(d, "dim")));
  if (d.count("ftrl_batch")) p.ftrl_batch = std::max(1, std::atoi(d["ftrl_batch"].c_str()));
  if (d.count("load_path")) load_path = d["load_path"];
}

}  // namespace admm
