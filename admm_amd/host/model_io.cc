// Text formats of the worker / global state files (src/admm.h:20-110,135-163): default iostream
// float formatting, sparse `index:value` lines.
#include <cmath>
#include <fstream>
#include <sstream>

#include "admm_host.h"

using admm::DeviceVector;
using admm::Runtime;

void LocalModel::InitModel(std::size_t fdim, ::admm::ArgParser &arg_parser) {
  // src/admm.h:112-128: rank r takes the r-th `psid : num_part` line
  worker_processor_.InitWorker(fdim);
  const int rank = Runtime::Get().GetRank();
  ADMM_CHECK((size_t)rank < arg_parser.psid.size());
  int sum_part = 0, pre_sum_part = 0;
  for (size_t i = 0; i < arg_parser.psid.size(); ++i) {
    sum_part += arg_parser.num_part[i];
    if (sum_part > rank) break;
    pre_sum_part = sum_part;
  }
  worker_processor_.psid_ = arg_parser.psid[rank];
  worker_processor_.num_part_ = arg_parser.num_part[rank];
  worker_processor_.partid_ = rank - pre_sum_part;
}

namespace {

void write_sparse(std::ostream &os, const std::vector<float> &v) {
  os << v.size() << '\n';
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] != 0) os << i << ':' << v[i] << '\n';
}

}  // namespace

// admm_weight_<psid>: five sections (x, bias, dual, n, z), each `size` then `i:value` for the
// non-zero entries (src/admm.h:83-110)
void LocalModel::SaveState(const std::string &path) const {
  std::ofstream os(path);
  ADMM_CHECK(os.is_open());
  const ::admm::Worker &w = worker_processor_;
  for (const DeviceVector *v : {&w.base_vec_, &w.bias_vec_, &w.langr_vec_, &w.nt_vec_, &w.zt_vec_})
    write_sparse(os, v->to_host());
}

// <psid>.bin: `-1:<bias weight>` then `i-1:value` where |x_i + bias_i| > e^-6 (src/admm.h:72-81)
void LocalModel::Save(const std::string &path) const {
  std::ofstream os(path);
  ADMM_CHECK(os.is_open());
  const std::vector<float> x = worker_processor_.base_vec_.to_host();
  const std::vector<float> b = worker_processor_.bias_vec_.to_host();
  if (x.empty()) return;
  os << -1 << ":" << x[0] << '\n';
  const float thr = std::exp(-6.0f);
  for (size_t i = 1; i < x.size(); ++i) {
    const float val = x[i] + b[i];
    if (val > thr || val < -thr) os << i - 1 << ":" << val << '\n';
  }
}

// src/admm.h:20-70
bool LocalModel::Load(const std::string &path) {
  std::ifstream is(path);
  if (!is.is_open()) return false;
  std::vector<float> sec[5];
  std::string line;
  int index = 0;
  while (std::getline(is, line)) {
    if (line.empty()) continue;
    const size_t loc = line.find(':');
    if (loc == std::string::npos) {
      ++index;
      if (index >= 1 && index <= 5) sec[index - 1].assign((size_t)std::stoi(line), 0.0f);
      continue;
    }
    if (index >= 1 && index <= 5) {
      const size_t i = (size_t)std::stoi(line.substr(0, loc));
      ADMM_CHECK(i < sec[index - 1].size());
      sec[index - 1][i] = std::stof(line.substr(loc + 1));
    }
  }
  ::admm::Worker &w = worker_processor_;
  w.base_vec_.from_host(sec[0]);
  w.bias_vec_.from_host(sec[1]);
  w.langr_vec_.from_host(sec[2]);
  w.nt_vec_.from_host(sec[3]);
  w.zt_vec_.from_host(sec[4]);
  return true;
}

// global_params_<psid>: step_size, global_var, bias_var, dim, then sparse weights (src/admm.h:153-163)
void GlobalModel::Save(const std::string &path) const {
  std::ofstream os(path);
  ADMM_CHECK(os.is_open());
  os << admm_params_.step_size << '\n';
  os << admm_params_.global_var << '\n';
  os << admm_params_.bias_var << '\n';
  os << admm_params_.dim << '\n';
  const std::vector<float> z = admm_params_.global_weights.to_host();
  for (size_t i = 0; i < admm_params_.dim && i < z.size(); ++i)
    if (z[i] != 0) os << i << ':' << z[i] << '\n';
}

// src/admm.h:135-152
bool GlobalModel::Load(const std::string &path) {
  std::ifstream is(path);
  if (!is.is_open()) return false;
  is >> admm_params_.step_size >> admm_params_.global_var >> admm_params_.bias_var >> admm_params_.dim;
  std::vector<float> z(admm_params_.dim, 0.0f);
  std::string index, value;
  while (std::getline(is, index, ':') && std::getline(is, value, '\n')) {
    const size_t i = (size_t)std::atoi(index.c_str());
    ADMM_CHECK(i < z.size());
    z[i] = (float)std::atof(value.c_str());
  }
  admm_params_.global_weights.from_host(z);
  return true;
}
