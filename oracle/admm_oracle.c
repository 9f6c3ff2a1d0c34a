/*
 * admm_oracle.c -- CPU restatement of the ADMM / FTRL sparse logistic-regression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under admm_amd/ links, loads or calls this file; it exists
 * so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have an independent
 * checker for the HIP kernels.
 *
 * Each function follows the reference source it names (paths relative to /root/reference).
 * The reference cannot be compiled here: every file of src/ includes dmlc-core and rabit
 * headers whose sources are absent (empty third_party/ submodules, no network), and building
 * against stand-in headers is not a reference build.  So this restatement is pinned only by
 *   (1) the known answers the reference's own sources/tests imply (test/wtest.libsvm,
 *       test/ftrl_test.cc:37 "prediction 0.5 at zero weights"), and
 *   (2) nine FTRL state values and the zero-weight logloss recorded from a run of the reference
 *       sources on test/wtest.libsvm (tests/golden/wtest_known_answers.json).
 * Beyond those cases: PARITY UNPINNED.
 *
 * Arithmetic conventions of the reference that are reproduced on purpose:
 *   - storage is float; sqrt/exp/log on a float argument are the C double functions, so the
 *     per-element intermediates are double and are rounded to float on store;
 *   - dot products (dmlc Row::SDot) and the logloss / AUC accumulators are float and strictly
 *     sequential;
 *   - margins are clamped to [-35, 35] before the sigmoid.
 */
#include "admm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Sample set: bias prepend + index shift                                                      */
/* ------------------------------------------------------------------------------------------ */

/* src/sample_set.cc:70-88 (SampleSet::TranslateData): every row gets (index 0, value 1) in
 * front and all original indices move up by one. */
void orc_translate(const uint64_t *rowptr, const uint32_t *col, const float *val, uint64_t nrows,
                   uint64_t *rowptr_out, uint32_t *col_out, float *val_out) {
  uint64_t o = 0;
  for (uint64_t r = 0; r < nrows; ++r) {
    rowptr_out[r] = o;
    col_out[o] = 0;
    val_out[o] = 1.0f;
    ++o;
    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      col_out[o] = col[k] + 1u;
      val_out[o] = val ? val[k] : 1.0f;
      ++o;
    }
  }
  rowptr_out[nrows] = o;
}

/* dmlc::Row::SDot as the reference uses it (src/ftrl.cc:69, src/metrics.cc:54,100): float
 * accumulator, elements in storage order, value==NULL means implicit 1. */
static float sdot(const uint32_t *idx, const float *val, uint64_t len, const float *w) {
  float sum = 0.0f;
  if (val) {
    for (uint64_t i = 0; i < len; ++i) sum += w[idx[i]] * val[i];
  } else {
    for (uint64_t i = 0; i < len; ++i) sum += w[idx[i]];
  }
  return sum;
}

static float clamp35(float m) {
  /* std::max(std::min(m, 35.f), -35.f): src/ftrl.cc:72, src/metrics.cc:58,104 */
  float a = m < 35.0f ? m : 35.0f;
  return a > -35.0f ? a : -35.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* FTRL-proximal                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* The lazy closed-form weight of src/ftrl.cc:61-66.  The threshold compares against l1 but the
 * shrink term is the literal 1 (sign * 1), exactly as the reference has it. */
static float ftrl_weight(float z, float n, float c, float alpha, float beta, float l1, float l2) {
  float sign = z < 0.0f ? -1.0f : 1.0f;
  if (sign * z <= l1) return 0.0f;
  float num = sign * 1 - z + l2 * c;
  double den = ((double)beta + sqrt((double)n)) / (double)alpha + (double)l2;
  return (float)((double)num / den);
}

/* src/ftrl.cc:54-73 (FtrlSolver::Predict) */
float orc_ftrl_predict(const uint32_t *idx, const float *val, uint64_t len, float *w,
                       const float *z, const float *n, const float *offset,
                       const float *reg_offset, float alpha, float beta, float l1, float l2) {
  for (uint64_t i = 0; i < len; ++i) {
    uint32_t j = idx[i];
    w[j] = ftrl_weight(z[j], n[j], reg_offset[j], alpha, beta, l1, l2);
  }
  float inner = sdot(idx, val, len, w) + sdot(idx, val, len, offset);
  return (float)(1.0 / (1 + exp(-(double)clamp35(inner))));
}

/* src/ftrl.cc:75-88 (FtrlSolver::Update).  The gradient is (p - y) for every stored coordinate,
 * feature values are not used (src/ftrl.cc:78-81). */
void orc_ftrl_update(float predict, float label, const uint32_t *idx, uint64_t len,
                     const float *w, float *z, float *n, float alpha) {
  int y = label > 0 ? 1 : 0;
  float loss = predict - y;
  for (uint64_t i = 0; i < len; ++i) {
    uint32_t j = idx[i];
    double sigma = (sqrt((double)(n[j] + loss * loss)) - sqrt((double)n[j])) / (double)alpha;
    z[j] = (float)((double)z[j] + ((double)loss - sigma * (double)w[j]));
    n[j] = n[j] + loss * loss;
  }
}

/* src/ftrl.cc:101-145 (FtrlSolver::Run): one sequential pass, then one more Predict on the last
 * row so that its coordinates carry post-update weights.  Rows are already bias-translated. */
void orc_ftrl_run(const uint64_t *rowptr, const uint32_t *col, const float *val,
                  const float *label, uint64_t nrows, float *w, float *z, float *n,
                  const float *offset, const float *reg_offset, float alpha, float beta, float l1,
                  float l2) {
  for (uint64_t r = 0; r < nrows; ++r) {
    uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
    const float *v = val ? val + b : NULL;
    float p = orc_ftrl_predict(col + b, v, len, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
    orc_ftrl_update(p, label[r], col + b, len, w, z, n, alpha);
  }
  if (nrows > 0) {
    uint64_t b = rowptr[nrows - 1], len = rowptr[nrows] - b;
    const float *v = val ? val + b : NULL;
    (void)orc_ftrl_predict(col + b, v, len, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
  }
}

/* Mini-batch generalisation (not in the reference; batch == 1 is src/ftrl.cc:101-145 again).
 * All rows of a batch are predicted from the state at the start of the batch; then every
 * coordinate touched in the batch receives the sum of its per-row updates.  Because the weight
 * is constant inside a batch the per-row sigmas telescope:
 *   n' = n + sum g^2 ,  z' = z + sum g - w * (sqrt(n') - sqrt(n)) / alpha .
 * scratch_g1 / scratch_g2 are dim doubles, zero on entry and on exit. */
void orc_ftrl_run_batch(const uint64_t *rowptr, const uint32_t *col, const float *val,
                        const float *label, uint64_t nrows, float *w, float *z, float *n,
                        const float *offset, const float *reg_offset, float alpha, float beta,
                        float l1, float l2, uint64_t batch, double *scratch_g1,
                        double *scratch_g2) {
  if (batch == 0) batch = 1;
  for (uint64_t r0 = 0; r0 < nrows; r0 += batch) {
    uint64_t r1 = r0 + batch < nrows ? r0 + batch : nrows;
    for (uint64_t r = r0; r < r1; ++r) {
      uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
      const float *v = val ? val + b : NULL;
      float p =
          orc_ftrl_predict(col + b, v, len, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
      int y = label[r] > 0 ? 1 : 0;
      float g = p - y;
      for (uint64_t k = b; k < b + len; ++k) {
        scratch_g1[col[k]] += (double)g;
        scratch_g2[col[k]] += (double)(g * g);
      }
    }
    for (uint64_t r = r0; r < r1; ++r) {
      for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        uint32_t j = col[k];
        if (scratch_g2[j] == 0.0 && scratch_g1[j] == 0.0) continue; /* already applied */
        float n_new = (float)((double)n[j] + scratch_g2[j]);
        double sigma = (sqrt((double)n_new) - sqrt((double)n[j])) / (double)alpha;
        z[j] = (float)((double)z[j] + (scratch_g1[j] - sigma * (double)w[j]));
        n[j] = n_new;
        scratch_g1[j] = 0.0;
        scratch_g2[j] = 0.0;
      }
    }
  }
  if (nrows > 0) {
    uint64_t b = rowptr[nrows - 1], len = rowptr[nrows] - b;
    const float *v = val ? val + b : NULL;
    (void)orc_ftrl_predict(col + b, v, len, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Metrics                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static float margin_groups(const uint32_t *idx, const float *val, uint64_t len,
                           const float *const *groups, int ngroups) {
  /* src/metrics.cc:54-57 / :100-103: float sum over weight groups, group 0 first */
  float inner = sdot(idx, val, len, groups[0]);
  for (int g = 1; g < ngroups; ++g) inner += sdot(idx, val, len, groups[g]);
  return inner;
}

/* src/metrics.cc:91-117 (Metrics::LogLoss): float running sum, double per-row term. */
float orc_logloss(const uint64_t *rowptr, const uint32_t *col, const float *val,
                  const float *label, uint64_t nrows, const float *const *groups, int ngroups) {
  float sum = 0;
  int count = 0;
  for (uint64_t r = 0; r < nrows; ++r) {
    uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
    float inner = margin_groups(col + b, val ? val + b : NULL, len, groups, ngroups);
    double predict = 1.0f / (1 + exp(-(double)clamp35(inner)));
    sum = (float)((double)sum + (label[r] > 0 ? -log(predict) : -log(1 - predict)));
    count++;
  }
  return sum / count;
}

/* The same quantity with a double accumulator and no intermediate float rounding of the sum:
 * what a parallel reduction is compared with at row counts where the float running sum of the
 * reference has itself lost digits.  Returns the SUM (not the mean). */
double orc_logloss_sum_f64(const uint64_t *rowptr, const uint32_t *col, const float *val,
                           const float *label, uint64_t nrows, const float *const *groups,
                           int ngroups, float *out_p) {
  double sum = 0;
  for (uint64_t r = 0; r < nrows; ++r) {
    uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
    float inner = margin_groups(col + b, val ? val + b : NULL, len, groups, ngroups);
    double predict = 1.0f / (1 + exp(-(double)clamp35(inner)));
    if (out_p) out_p[r] = (float)predict;
    sum += (label[r] > 0 ? -log(predict) : -log(1 - predict));
  }
  return sum;
}

typedef struct {
  float v;
  int i;
} rank_item;

static int rank_cmp(const void *a, const void *b) {
  const rank_item *x = (const rank_item *)a, *y = (const rank_item *)b;
  if (x->v < y->v) return -1;
  if (y->v < x->v) return 1;
  /* std::stable_sort keeps equal keys in input order */
  return (x->i > y->i) - (x->i < y->i);
}

/* src/metrics.cc:14-43 (Metrics::Sort): 1-based average ranks of tied groups, as floats. */
void orc_tied_rank(const float *input, int dim, float *output) {
  if (dim <= 0) return;
  rank_item *items = (rank_item *)malloc(sizeof(rank_item) * (size_t)dim);
  for (int i = 0; i < dim; ++i) {
    items[i].v = input[i];
    items[i].i = i;
  }
  qsort(items, (size_t)dim, sizeof(rank_item), rank_cmp);
  int last_rank = 0;
  float cur = items[0].v;
  for (int i = 0; i < dim; ++i) {
    if (cur != items[i].v) {
      cur = items[i].v;
      for (int j = last_rank; j < i; ++j) output[items[j].i] = (last_rank + i + 1.0f) / 2;
      last_rank = i;
    }
    if (i == dim - 1) {
      for (int j = last_rank; j < i + 1; ++j) output[items[j].i] = (last_rank + i + 2.0f) / 2;
    }
  }
  free(items);
}

/* Rank-sum AUC over already computed scores: src/metrics.cc:66-80.  Positives are the rows whose
 * label truncated to int equals 1 (src/metrics.cc:60,72); all accumulators are float. */
float orc_auc_from_scores(const float *scores, const float *label, int nrows) {
  float *ranks = (float *)malloc(sizeof(float) * (size_t)(nrows > 0 ? nrows : 1));
  orc_tied_rank(scores, nrows, ranks);
  float positive_sum = 0, total_p = 0, total_n = 0;
  for (int i = 0; i < nrows; ++i) {
    if ((int)label[i] == 1) {
      total_p++;
      positive_sum += ranks[i];
    } else {
      total_n++;
    }
  }
  free(ranks);
  return positive_sum / (total_p * total_n) - (total_p + 1) / (2 * total_n);
}

/* The same statistic in double (tie-aware), for row counts where float ranks cannot be exact. */
double orc_auc_from_scores_f64(const float *scores, const float *label, int nrows) {
  if (nrows <= 0) return NAN;
  rank_item *items = (rank_item *)malloc(sizeof(rank_item) * (size_t)nrows);
  for (int i = 0; i < nrows; ++i) {
    items[i].v = scores[i];
    items[i].i = i;
  }
  qsort(items, (size_t)nrows, sizeof(rank_item), rank_cmp);
  double positive_sum = 0, total_p = 0, total_n = 0;
  int s = 0;
  while (s < nrows) {
    int e = s;
    while (e < nrows && items[e].v == items[s].v) ++e;
    double avg = ((double)s + (double)e + 1.0) / 2.0; /* mean of 1-based ranks s+1..e */
    for (int k = s; k < e; ++k) {
      if ((int)label[items[k].i] == 1) {
        total_p += 1;
        positive_sum += avg;
      } else {
        total_n += 1;
      }
    }
    s = e;
  }
  free(items);
  return positive_sum / (total_p * total_n) - (total_p + 1) / (2 * total_n);
}

/* src/metrics.cc:45-89 (Metrics::Auc, single-model form) */
float orc_auc(const uint64_t *rowptr, const uint32_t *col, const float *val, const float *label,
              uint64_t nrows, const float *const *groups, int ngroups) {
  float *scores = (float *)malloc(sizeof(float) * (size_t)(nrows > 0 ? nrows : 1));
  for (uint64_t r = 0; r < nrows; ++r) {
    uint64_t b = rowptr[r], len = rowptr[r + 1] - b;
    float inner = margin_groups(col + b, val ? val + b : NULL, len, groups, ngroups);
    double predict = 1.0f / (1 + exp(-(double)clamp35(inner)));
    scores[r] = (float)predict;
  }
  float auc = orc_auc_from_scores(scores, label, (int)nrows);
  free(scores);
  return auc;
}

/* ------------------------------------------------------------------------------------------ */
/* ADMM algebra                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* src/workers.cc:121-125 (Worker::GetWeights): payload_i = x_i * rho + y_i */
void orc_payload(const float *x, const float *y, float rho, float *out, uint64_t dim) {
  for (uint64_t i = 0; i < dim; ++i) out[i] = x[i] * rho + y[i];
}

/* src/master.cc:5-20 (Master::GlobalUpdate), live branch: z_i = S_i / (N * rho + global_var).
 * With l1 > 0 the sum is soft-thresholded by l1 first (elastic-net consensus; l1 == 0 is the
 * reference exactly).  The branch the reference keeps commented out (src/master.cc:10-16) is
 * orc_global_update_commented below. */
void orc_global_update(const float *sum, int num_worker, float rho, float global_var, float l1,
                       float *zg, uint64_t dim) {
  for (uint64_t i = 0; i < dim; ++i) {
    float s = sum[i];
    if (l1 > 0.0f) {
      if (s > l1) s = s - l1;
      else if (s < -l1) s = s + l1;
      else s = 0.0f;
    }
    zg[i] = s / (num_worker * rho + global_var);
  }
}

/* src/master.cc:7,10-16: crit = global_var / rho; z = (S -+ crit) / N outside [-crit, crit]. */
void orc_global_update_commented(const float *sum, int num_worker, float rho, float global_var,
                                 float *zg, uint64_t dim) {
  float crit = global_var / rho;
  for (uint64_t i = 0; i < dim; ++i) {
    if (sum[i] > crit) zg[i] = (sum[i] - crit) / num_worker;
    else if (sum[i] < -crit) zg[i] = (sum[i] + crit) / num_worker;
    else zg[i] = 0;
  }
}

/* src/workers.cc:115-119 (Worker::LangrangeUpdate): y_i += rho * (x_i - z_i) */
void orc_dual_update(float *y, const float *x, const float *zg, float rho, uint64_t dim) {
  for (uint64_t i = 0; i < dim; ++i) y[i] += rho * (x[i] - zg[i]);
}

/* Residual norms defined by this build (the reference prints none):
 * r2 = sum (x - z)^2, s2 = rho^2 * sum (z - z_prev)^2, both in double. */
void orc_residuals(const float *x, const float *zg, const float *zg_prev, float rho,
                   uint64_t dim, double *r2, double *s2) {
  double a = 0, b = 0;
  for (uint64_t i = 0; i < dim; ++i) {
    double d = (double)x[i] - (double)zg[i];
    double e = (double)zg[i] - (double)zg_prev[i];
    a += d * d;
    b += e * e;
  }
  *r2 = a;
  *s2 = (double)rho * (double)rho * b;
}

/* ------------------------------------------------------------------------------------------ */
/* Batch local objective (no counterpart in the reference)                                     */
/* ------------------------------------------------------------------------------------------ */

/* f(x) = sum_rows [ softplus(m_r) - t_r * m_r ] + y^T (x - z) + rho/2 * ||x - z||^2 ,
 * m = X x + X offset , t = (label > 0) , g = X^T (sigmoid(m) - t) + y + rho (x - z).
 * Margins are not clamped here so that f and g stay consistent for the line search.
 * Inputs are float, every accumulation is double.  gbuf is dim doubles (output). */
double orc_logistic_fg(const uint64_t *rowptr, const uint32_t *col, const float *val,
                       const float *label, uint64_t nrows, uint64_t dim, const float *x,
                       const float *offset, const float *zg, const float *ydual, float rho,
                       double *gbuf) {
  double f = 0;
  for (uint64_t i = 0; i < dim; ++i) gbuf[i] = 0;
  for (uint64_t r = 0; r < nrows; ++r) {
    double m = 0;
    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      double v = val ? (double)val[k] : 1.0;
      m += ((double)x[col[k]] + (offset ? (double)offset[col[k]] : 0.0)) * v;
    }
    double t = label[r] > 0 ? 1.0 : 0.0;
    double sp = (m > 0 ? m : 0) + log1p(exp(-fabs(m)));
    f += sp - t * m;
    double res = 1.0 / (1.0 + exp(-m)) - t;
    for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      double v = val ? (double)val[k] : 1.0;
      gbuf[col[k]] += res * v;
    }
  }
  for (uint64_t i = 0; i < dim; ++i) {
    double d = (double)x[i] - (zg ? (double)zg[i] : 0.0);
    double yd = ydual ? (double)ydual[i] : 0.0;
    f += yd * d + 0.5 * (double)rho * d * d;
    gbuf[i] += yd + (double)rho * d;
  }
  return f;
}

/* ------------------------------------------------------------------------------------------ */
/* libsvm text -> CSR (dmlc-core LibSVMParser + InputSplit are absent: PARITY UNPINNED)        */
/* ------------------------------------------------------------------------------------------ */

/* Byte range of part `part` of `nparts`: rows whose first byte lies in
 * [part*step, (part+1)*step), step = ceil(len/nparts).  Returns begin/end offsets aligned to
 * line starts. */
void orc_part_range(const char *text, uint64_t len, uint32_t part, uint32_t nparts,
                    uint64_t *begin, uint64_t *end) {
  uint64_t step = (len + nparts - 1) / nparts;
  uint64_t b = (uint64_t)part * step, e = (uint64_t)(part + 1) * step;
  if (b > len) b = len;
  if (e > len) e = len;
  while (b > 0 && b < len && text[b - 1] != '\n') ++b;
  while (e > 0 && e < len && text[e - 1] != '\n') ++e;
  *begin = b;
  *end = e;
}

/* Two-call protocol: with col == NULL only counts rows and nnz.  A row is
 * `label [idx[:val]]...`; blank lines are skipped; `idx` alone means value 1. */
int orc_parse_libsvm(const char *text, uint64_t begin, uint64_t end, uint64_t *nrows_out,
                     uint64_t *nnz_out, uint64_t *rowptr, uint32_t *col, float *val,
                     float *label) {
  uint64_t nrows = 0, nnz = 0;
  uint64_t p = begin;
  while (p < end) {
    uint64_t q = p;
    while (q < end && text[q] != '\n') ++q;
    /* line = [p, q) */
    uint64_t s = p;
    while (s < q && (text[s] == ' ' || text[s] == '\t' || text[s] == '\r')) ++s;
    if (s < q) {
      char buf[64];
      uint64_t t = s;
      while (t < q && text[t] != ' ' && text[t] != '\t' && text[t] != '\r') ++t;
      uint64_t l = t - s < 63 ? t - s : 63;
      memcpy(buf, text + s, l);
      buf[l] = 0;
      if (col) {
        rowptr[nrows] = nnz;
        label[nrows] = strtof(buf, NULL);
      }
      s = t;
      while (s < q) {
        while (s < q && (text[s] == ' ' || text[s] == '\t' || text[s] == '\r')) ++s;
        if (s >= q) break;
        t = s;
        while (t < q && text[t] != ' ' && text[t] != '\t' && text[t] != '\r') ++t;
        l = t - s < 63 ? t - s : 63;
        memcpy(buf, text + s, l);
        buf[l] = 0;
        char *colon = strchr(buf, ':');
        if (col) {
          if (colon) {
            *colon = 0;
            col[nnz] = (uint32_t)strtoull(buf, NULL, 10);
            val[nnz] = strtof(colon + 1, NULL);
          } else {
            col[nnz] = (uint32_t)strtoull(buf, NULL, 10);
            val[nnz] = 1.0f;
          }
        }
        ++nnz;
        s = t;
      }
      ++nrows;
    }
    p = q + 1;
  }
  if (col) rowptr[nrows] = nnz;
  *nrows_out = nrows;
  *nnz_out = nnz;
  return 0;
}
