// FtrlSolver, Worker, Master and Metrics on top of the C ABI.
#include <cmath>

#include <cstdlib>

#include "admm_host.h"

using admm::DeviceVector;
using admm::Runtime;

namespace ftrl {

FtrlSolver::FtrlSolver(real_t lambda_1, real_t lambda_2, real_t alpha_init, real_t beta_init, std::size_t dim_init)
    : l_1_(lambda_1), l_2_(lambda_2), alpha_(alpha_init), beta_(beta_init), dim_(dim_init) {
  weight_.resize(dim_);
  mid_weight_.resize(dim_);
  squared_sum_.resize(dim_);
}

void FtrlSolver::Init(::admm::FtrlConfig &params) {
  l_1_ = params.l_1;
  l_2_ = params.l_2;
  alpha_ = params.alpha;
  beta_ = params.beta;
  dim_ = params.dim;
  batch_ = params.ftrl_batch;
  weight_.resize(dim_);  // zero-filled (src/ftrl.cc:43-51)
  mid_weight_.resize(dim_);
  squared_sum_.resize(dim_);
}

void FtrlSolver::Assign(const DeviceVector &x, const DeviceVector &y) {
  mid_weight_ = x;
  squared_sum_ = y;
}

// src/ftrl.cc:54-73
FtrlSolver::real_t FtrlSolver::Predict(::admm::Row &x, const DeviceVector &offset, const DeviceVector &reg_offset) {
  ADMM_CHECK(offset.size() == dim_ && reg_offset.size() == dim_);
  std::vector<uint32_t> index(x.length);
  for (size_t i = 0; i < x.length; ++i) {
    ADMM_CHECK(x.index[i] < dim_);
    index[i] = (uint32_t)x.index[i];
  }
  float p = 0;
  ADMM_CALL(admm_ftrl_predict(::admm::Runtime::Get().ctx(), index.data(), x.value, (uint32_t)x.length, weight_.data(),
                              mid_weight_.data(), squared_sum_.data(), offset.data(), reg_offset.data(), alpha_, beta_,
                              l_1_, l_2_, &p));
  return p;
}

// src/ftrl.cc:75-88
void FtrlSolver::Update(real_t predict, const ::admm::Row &x, const DeviceVector &) {
  std::vector<uint32_t> index(x.length);
  for (size_t i = 0; i < x.length; ++i) {
    ADMM_CHECK(x.index[i] < dim_);
    index[i] = (uint32_t)x.index[i];
  }
  ADMM_CALL(admm_ftrl_update(::admm::Runtime::Get().ctx(), index.data(), (uint32_t)x.length, x.label, predict,
                             weight_.data(), mid_weight_.data(), squared_sum_.data(), alpha_));
}

void FtrlSolver::Run(SampleSet &train_set, SampleSet &, const DeviceVector &offset, const DeviceVector &reg_offset) {
  ADMM_CHECK(train_set.device() != nullptr);
  ADMM_CHECK(offset.size() == dim_ && reg_offset.size() == dim_);
  ADMM_CALL(admm_ftrl_pass(train_set.device(), weight_.data(), mid_weight_.data(), squared_sum_.data(),
                           offset.data(), reg_offset.data(), alpha_, beta_, l_1_, l_2_, batch_));
}

}  // namespace ftrl

namespace admm {

void Worker::InitWorker(std::size_t fdim) {
  base_vec_.resize(fdim);
  bias_vec_.resize(fdim);
  langr_vec_.resize(fdim);
  nt_vec_.resize(fdim);
  zt_vec_.resize(fdim);
  load = false;
}

// the worker's data parts: loaded on the first call, resident afterwards
void Worker::LoadParts(const AdmmConfig &admm_params) {
  if (parts_.empty()) {
    const std::string train_name = admm_params.train_path + "/" + psid_ + "_aggregated/part-00000";
    // The batch objective of the lbfgs solver is a sum over all rows, whatever the file's partition:
    // its num_part pieces are loaded as one resident set (the reference splits a file into parts to
    // bound the host memory of one load, src/workers.cc:73-77; HBM holds the whole shard).
    const int nparts = admm_params.local_solver == "lbfgs" ? 1 : num_part_;
    for (int part = 0; part < nparts; ++part) {
      parts_.emplace_back(new SampleSet);
      ADMM_CHECK(parts_.back()->Initialize(train_name, part, nparts, admm_params.dim));
      if (admm_params.local_solver == "lbfgs") {
        // ADMM_LEAN_SETS=1: keep only the forms the solver's sweeps read (about 1.4 instead of 3 times the CSR in HBM)
        const char *lean = std::getenv("ADMM_LEAN_SETS");
        parts_.back()->PrepareTransposed(0, 16384, true, lean && lean[0] == '1');
      }
      else if (admm_params.ftrl_batch > 1)
        parts_.back()->PrepareTransposed(admm_params.ftrl_batch, 0, true);
    }
  }
}

// The per-worker bias model (src/workers.cc:89-113; the reference's main loop has its call commented out,
// src/admm_allreduce.cpp:65): one FTRL pass from zero state with l_2 = 200, l_1 = 1000, the base model as
// the fixed offset and no proximal centre; bias_vec_ <- the pass's weights.  It reads the worker's
// resident parts (the reference builds its file name without the '/' after train_path, :101, so there the
// call only finds its data when train_path ends in one).
void Worker::BiasUpdate(SampleSet &, SampleSet &test_set, const AdmmConfig &admm_params) {
  ADMM_CHECK(admm_params.local_solver != "lbfgs");  // the bias model is an FTRL model
  LoadParts(admm_params);
  FtrlConfig ftrl_params(admm_params);
  ftrl_params.l_2 = 200;
  ftrl_params.l_1 = 1000;
  ftrl_params.ftrl_batch = admm_params.ftrl_batch;
  ::ftrl::FtrlSolver ftrl_processor;
  ftrl_processor.Init(ftrl_params);
  DeviceVector reg_offset(base_vec_.size());  // zero-filled
  for (auto &part : parts_) ftrl_processor.Run(*part, test_set, base_vec_, reg_offset);
  bias_vec_ = ftrl_processor.weight_;
}

// The x-update (src/workers.cc:51-87).  Parts are loaded on the first call and stay resident.
void Worker::BaseUpdate(SampleSet &, SampleSet &test_set, const AdmmConfig &admm_params) {
  LoadParts(admm_params);
  if (admm_params.local_solver == "lbfgs") {
    // batch objective: logistic loss + y.(x - z) + rho/2 |x - z|^2, warm-started at the last x
    ADMM_CHECK(parts_.size() == 1);
    ADMM_CALL(admm_lbfgs_solve(parts_[0]->device(), base_vec_.data(), bias_vec_.data(),
                               admm_params.global_weights.data(), langr_vec_.data(), admm_params.step_size,
                               &admm_params.lbfgs, &last_solve_));
    return;
  }
  FtrlConfig ftrl_params(admm_params);
  ftrl_params.l_2 = 200;  // src/workers.cc:53-54
  ftrl_params.l_1 = 1;
  ftrl_params.ftrl_batch = admm_params.ftrl_batch;
  ::ftrl::FtrlSolver ftrl_processor;
  ftrl_processor.Init(ftrl_params);
  ftrl_processor.squared_sum_ = nt_vec_;
  ftrl_processor.mid_weight_ = zt_vec_;
  for (auto &part : parts_)  // reg_offset = global weights; the dual term is not part of it (:59)
    ftrl_processor.Run(*part, test_set, bias_vec_, admm_params.global_weights);
  base_vec_ = ftrl_processor.weight_;
  if (load) {
    nt_vec_ = ftrl_processor.squared_sum_;
    zt_vec_ = ftrl_processor.mid_weight_;
    load = false;
  }
}

void Worker::LangrangeUpdate(const AdmmConfig &admm_params) {
  ADMM_CALL(admm_dual_update(Runtime::Get().ctx(), langr_vec_.data(), base_vec_.data(),
                             admm_params.global_weights.data(), admm_params.step_size, langr_vec_.size()));
}

void Worker::GetWeights(AdmmConfig &admm_params, DeviceVector &out) const {
  ADMM_CHECK(out.size() >= base_vec_.size());  // the main loop keeps a few floats of its own behind the payload
  ADMM_CALL(admm_payload(Runtime::Get().ctx(), base_vec_.data(), langr_vec_.data(), admm_params.step_size, out.data(),
                         base_vec_.size()));
}

bool Master::GlobalUpdate(const DeviceVector &workers, AdmmConfig &admm_params, int num_worker) {
  // (the main loop's buffer carries a few floats of its own behind the payload: the update covers the weights only)
  ADMM_CHECK(workers.size() >= admm_params.global_weights.size());
  return admm_global_update(Runtime::Get().ctx(), workers.data(), num_worker, admm_params.step_size,
                            admm_params.global_var, admm_params.global_l1, admm_params.global_weights.data(),
                            admm_params.global_weights.size()) == ADMM_OK;
}

bool Master::Consensus(const DeviceVector &workers, AdmmConfig &admm_params, int num_worker, Worker &worker, double *r2,
                       double *s2) {
  return admm_consensus(Runtime::Get().ctx(), workers.data(), num_worker, admm_params.step_size,
                        admm_params.global_var, admm_params.global_l1, worker.base_vec_.data(),
                        worker.langr_vec_.data(), admm_params.global_weights.data(), worker.base_vec_.size(), r2,
                        s2) == ADMM_OK;
}

}  // namespace admm

void Metrics::Both(::admm::SampleSet &sample_set, std::vector<const DeviceVector *> &weights, real_t *loss,
                   real_t *auc) {
  ADMM_CHECK(sample_set.device() != nullptr && !weights.empty() && weights.size() <= 4);
  const float *groups[4];
  for (size_t i = 0; i < weights.size(); ++i) groups[i] = weights[i]->data();
  double l = 0, a = 0;
  ADMM_CALL(admm_metrics(sample_set.device(), groups, (int)weights.size(), loss ? &l : nullptr, auc ? &a : nullptr,
                         nullptr));
  if (loss) *loss = (real_t)l;
  if (auc) *auc = (real_t)a;
}

void Metrics::AllRanks(::admm::Runtime &rt, ::admm::SampleSet &sample_set, std::vector<const DeviceVector *> &weights,
                       double *loss, double *auc, uint64_t *rows) {
  ADMM_CHECK(!weights.empty() && weights.size() <= 4);
  const int world = rt.GetWorldSize(), rank = rt.GetRank();
  const uint64_t n = sample_set.device() ? (uint64_t)sample_set.Size() : 0;
  // row counts and loss sums of all ranks; floats carry them exactly in two halves each
  double lsum = 0;
  const float *groups[4];
  for (size_t i = 0; i < weights.size(); ++i) groups[i] = weights[i]->data();
  if (n) {
    double l = 0;
    ADMM_CALL(admm_metrics(sample_set.device(), groups, (int)weights.size(), &l, nullptr, nullptr));
    lsum = l * (double)n;
  }
  std::vector<float> meta((size_t)world * 4, 0.0f);
  meta[rank * 4 + 0] = (float)(n >> 20);
  meta[rank * 4 + 1] = (float)(n & 0xFFFFFu);
  meta[rank * 4 + 2] = (float)lsum;
  meta[rank * 4 + 3] = (float)(lsum - (double)meta[rank * 4 + 2]);
  DeviceVector dmeta(meta.size());
  dmeta.from_host(meta);
  rt.AllreduceSum(dmeta.data(), meta.size());
  meta = dmeta.to_host();
  uint64_t total = 0, off = 0;
  double ltotal = 0;
  for (int r = 0; r < world; ++r) {
    const uint64_t c = ((uint64_t)meta[r * 4 + 0] << 20) + (uint64_t)meta[r * 4 + 1];
    if (r < rank) off += c;
    total += c;
    ltotal += (double)meta[r * 4 + 2] + (double)meta[r * 4 + 3];
  }
  DeviceVector buf(2 * std::max<uint64_t>(total, 1));  // zero-filled: [scores | labels]
  if (n) {
    ADMM_CALL(admm_predict(sample_set.device(), groups, (int)weights.size(), buf.data() + off));
    ADMM_CALL(admm_csr_copy_labels(sample_set.device(), buf.data() + std::max<uint64_t>(total, 1) + off));
  }
  rt.AllreduceSum(buf.data(), buf.size());
  double a = 0;
  ADMM_CALL(admm_auc_scores(rt.ctx(), buf.data(), buf.data() + std::max<uint64_t>(total, 1), total, &a));
  if (loss) *loss = total ? ltotal / (double)total : 0.0;
  if (auc) *auc = a;
  if (rows) *rows = total;
}

// src/metrics.cc:14-46
void Metrics::Sort(const std::vector<real_t> &input, std::vector<real_t> &output) {
  output.assign(input.size(), 0);
  if (input.empty()) return;
  DeviceVector scores(input.size()), ranks(input.size());
  scores.from_host(input);
  ADMM_CALL(admm_tied_ranks(::admm::Runtime::Get().ctx(), scores.data(), input.size(), ranks.data()));
  output = ranks.to_host();
}

Metrics::real_t Metrics::Auc(::admm::SampleSet &sample_set, std::vector<const DeviceVector *> &weights, bool) {
  real_t a = 0;
  Both(sample_set, weights, nullptr, &a);
  return a;
}

Metrics::real_t Metrics::LogLoss(::admm::SampleSet &sample_set, std::vector<const DeviceVector *> &weights, bool) {
  real_t l = 0;
  Both(sample_set, weights, &l, nullptr);
  return l;
}
