/* Test-only entry points of the MI355X build: NOT part of the product library.
 *
 * libmacr_hip.so (include/macr_hip.h) exports none of these; they exist in macr_amd/csrc/libmacr_hip_test.so, the same
 * sources compiled with -DMACR_TEST_ENTRY_POINTS, which only tests/ load (macr_amd/_lib.py::test_lib).  They make an
 * internal invariant of the bf16 candidate filter observable -- the reference has no counterpart (it scores in fp32:
 * macr_mf/model.py:199). */
#ifndef MACR_HIP_TEST_H
#define MACR_HIP_TEST_H
#include "macr_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* TEST-ONLY: the invariant the bf16 candidate filter rests on, made observable.  Writes, for U query rows and N item
 * rows (dev, fp32 [U][d] / [N][d]), the RAW product the bf16 kernels compute -- two-term bf16 splits, hi*hi + hi*lo +
 * lo*hi on v_mfma_f32_32x32x16_bf16 in the listing pass's instruction order -- to prod (dev) fp32[U][N], and to margin
 * (dev) fp32[U] the error margin the filter grants each query at this c (what it subtracts from her threshold).
 * tests/ assert |prod - fp32 fmaf chain| <= margin element-wise on adversarial operands for every d; no product code
 * calls this.  (The reference scores in fp32: macr_mf/model.py:199.) */
size_t macr_test_bf16_products_workspace_bytes(int d, int U, int N);
int macr_test_bf16_products(int d, int U, int N, const float *users, const float *items, float c,
                            float *prod, float *margin, void *workspace, size_t workspace_bytes, void *stream);

/* TEST-ONLY: the same for the listing pass macr_score_topk runs (k_score_stream_c: the epilogue in the operand copies --
 * item rows scaled by sig_i, the bias -c*sig_i as a three-term bf16 slab in the last MFMA, query rows scaled for
 * DIRECT_MINUS_BOTH).  Writes the SCORE that pass would list for every (query, item) pair of the given kind, scores (dev)
 * fp32[U][N], and the margin per query; tests/ assert |scores - fp32 score| <= margin element-wise.  sig_u / sig_i (dev)
 * fp32[U] / [N] as the kind needs them. */
size_t macr_test_bf16_scores_workspace_bytes(int d, int U, int N);
int macr_test_bf16_scores(int score_kind, int d, int U, int N, const float *users, const float *items,
                          const float *sig_u, const float *sig_i, float c, float *scores, float *margin,
                          void *workspace, size_t workspace_bytes, void *stream);

/* TEST-ONLY: the same for the fp16 filter (MACR_EVAL_FILTER_F16; k_score_stream_h: one fp16 number per operand, the bias
 * -c*sig_i (x sig_u for DIRECT_MINUS_BOTH) as six cross terms of two three-term fp16 splits in the slab).  Workspace:
 * macr_test_bf16_scores_workspace_bytes.  A query whose row leaves fp16's range gets margin = +inf. */
int macr_test_f16_scores(int score_kind, int d, int U, int N, const float *users, const float *items,
                         const float *sig_u, const float *sig_i, float c, float *scores, float *margin,
                         void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
