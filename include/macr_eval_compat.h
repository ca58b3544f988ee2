/* ============================================================================
 * macr_eval_compat.h -- the reference's OWN C ABI of its native evaluator,
 * served by the MI355X kernels.
 *
 * weitianxin/MACR binds exactly two C functions
 * (macr_lightgcn/evaluator/cpp/apt_evaluate_foldout.pyx:11-19):
 *     c_top_k_array_index   macr_lightgcn/evaluator/cpp/include/tools.h:24
 *     evaluate_foldout      macr_lightgcn/evaluator/cpp/include/evaluate_foldout.h:115-118
 * libmacr_eval_compat.so exports them with the SAME names, argument lists and
 * conventions -- host pointers owned by the caller, `int **ground_truths` as
 * an array of borrowed row pointers, `void` return, `thread_num` accepted --
 * so the .pyx (or any other binding of those two functions) links against it
 * unchanged: "relink, change nothing" (INTEGRATION.md B.1).
 *
 * Behind the symbols: the call stages its buffers to the current HIP device,
 * runs macr_topk_scores / macr_metrics_foldout (include/macr_hip.h) on the
 * default stream and copies the result back before it returns.  There is no
 * host fallback: without a usable device the call reports and fails.
 *
 * Differences a caller can observe, both inside what the reference leaves open:
 *   - tie order: score descending, then ascending column index
 *     (std::partial_sort_copy leaves it unspecified);
 *   - errors: the reference has no error channel.  Here a failed call fills
 *     its output with -1 (rankings) / NaN (results), prints one line to
 *     stderr and sets macr_eval_compat_status() (0 = last call succeeded);
 *   - any top_k, as in the reference (above 128 the ranking runs in rounds of 128 positions; positions past the
 *     number of columns hold -1); thread_num is ignored.
 * ==========================================================================*/
#ifndef MACR_EVAL_COMPAT_H
#define MACR_EVAL_COMPAT_H

#ifdef __cplusplus
extern "C" {
#endif

/* tools.h:24 -- per row of the row-major (rows_num, columns_num) matrix scores_pt,
 * the column indices of its top_k largest entries, best first, into
 * rankings_pt (rows_num, top_k).  -inf entries (masked train items,
 * utility/batch_test.py:129) rank last. */
void c_top_k_array_index(float *scores_pt, int columns_num, int rows_num, int top_k,
                         int thread_num, int *rankings_pt);

/* evaluate_foldout.h:115-118 -- results (users_num, 5*rank_len) laid out
 * [precision | recall | ap | ndcg | mrr] x rank_len prefixes per user
 * (evaluate_foldout.h:16-112).  ground_truths[u] points at
 * ground_truths_num[u] item ids (any order). */
void evaluate_foldout(int users_num, int *rankings, int rank_len,
                      int **ground_truths, int *ground_truths_num,
                      int thread_num, float *results);

/* 0 if the last call of this thread succeeded, else the negative MACR_E_* /
 * HIP-derived code; message on stderr and in macr_eval_compat_error(). */
int         macr_eval_compat_status(void);
const char *macr_eval_compat_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MACR_EVAL_COMPAT_H */
