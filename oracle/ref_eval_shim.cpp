// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Thin extern "C" door onto the reference's own header-only C++ evaluator so
// that the oracle and the HIP path can be checked against the reference code
// itself.  The headers are compiled from where they lie under /root/reference
// (see oracle/Makefile, target _ref); nothing from the reference is copied
// into this repository and the resulting oracle/_ref/libref_eval.so is
// git-ignored.
//
//   c_top_k_array_index : macr_lightgcn/evaluator/cpp/include/tools.h:24
//   evaluate_foldout    : macr_lightgcn/evaluator/cpp/include/evaluate_foldout.h:115
#include "macr_lightgcn/evaluator/cpp/include/tools.h"
#include "macr_lightgcn/evaluator/cpp/include/evaluate_foldout.h"

extern "C" {

void ref_top_k_array_index(float *scores, int columns_num, int rows_num,
                           int top_k, int thread_num, int *rankings) {
    c_top_k_array_index(scores, columns_num, rows_num, top_k, thread_num, rankings);
}

void ref_evaluate_foldout(int users_num, int *rankings, int rank_len,
                          int **ground_truths, int *ground_truths_num,
                          int thread_num, float *results) {
    evaluate_foldout(users_num, rankings, rank_len, ground_truths,
                     ground_truths_num, thread_num, results);
}

}  // extern "C"
