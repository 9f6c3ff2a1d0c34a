/* CPU restatement of the ADMM / FTRL hot path -- TEST INFRASTRUCTURE ONLY (see admm_oracle.c). */
#ifndef ADMM_ORACLE_H_
#define ADMM_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void orc_translate(const uint64_t *rowptr, const uint32_t *col, const float *val, uint64_t nrows,
                   uint64_t *rowptr_out, uint32_t *col_out, float *val_out);

float orc_ftrl_predict(const uint32_t *idx, const float *val, uint64_t len, float *w,
                       const float *z, const float *n, const float *offset,
                       const float *reg_offset, float alpha, float beta, float l1, float l2);
void orc_ftrl_update(float predict, float label, const uint32_t *idx, uint64_t len,
                     const float *w, float *z, float *n, float alpha);
void orc_ftrl_run(const uint64_t *rowptr, const uint32_t *col, const float *val,
                  const float *label, uint64_t nrows, float *w, float *z, float *n,
                  const float *offset, const float *reg_offset, float alpha, float beta, float l1,
                  float l2);
void orc_ftrl_run_batch(const uint64_t *rowptr, const uint32_t *col, const float *val,
                        const float *label, uint64_t nrows, float *w, float *z, float *n,
                        const float *offset, const float *reg_offset, float alpha, float beta,
                        float l1, float l2, uint64_t batch, double *scratch_g1,
                        double *scratch_g2);

float orc_logloss(const uint64_t *rowptr, const uint32_t *col, const float *val,
                  const float *label, uint64_t nrows, const float *const *groups, int ngroups);
double orc_logloss_sum_f64(const uint64_t *rowptr, const uint32_t *col, const float *val,
                           const float *label, uint64_t nrows, const float *const *groups,
                           int ngroups, float *out_p);
void orc_tied_rank(const float *input, int dim, float *output);
float orc_auc_from_scores(const float *scores, const float *label, int nrows);
double orc_auc_from_scores_f64(const float *scores, const float *label, int nrows);
float orc_auc(const uint64_t *rowptr, const uint32_t *col, const float *val, const float *label,
              uint64_t nrows, const float *const *groups, int ngroups);

void orc_payload(const float *x, const float *y, float rho, float *out, uint64_t dim);
void orc_global_update(const float *sum, int num_worker, float rho, float global_var, float l1,
                       float *zg, uint64_t dim);
void orc_global_update_commented(const float *sum, int num_worker, float rho, float global_var,
                                 float *zg, uint64_t dim);
void orc_dual_update(float *y, const float *x, const float *zg, float rho, uint64_t dim);
void orc_residuals(const float *x, const float *zg, const float *zg_prev, float rho,
                   uint64_t dim, double *r2, double *s2);

double orc_logistic_fg(const uint64_t *rowptr, const uint32_t *col, const float *val,
                       const float *label, uint64_t nrows, uint64_t dim, const float *x,
                       const float *offset, const float *zg, const float *ydual, float rho,
                       double *gbuf);

void orc_part_range(const char *text, uint64_t len, uint32_t part, uint32_t nparts,
                    uint64_t *begin, uint64_t *end);
int orc_parse_libsvm(const char *text, uint64_t begin, uint64_t end, uint64_t *nrows_out,
                     uint64_t *nnz_out, uint64_t *rowptr, uint32_t *col, float *val,
                     float *label);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_ORACLE_H_ */
