// libmacr_eval_compat.so: the reference's two native evaluator entry points (tools.h:24, evaluate_foldout.h:115-118;
// bound by apt_evaluate_foldout.pyx:11-19) with their original signatures, staged onto the HIP kernels of
// libmacr_hip.so.  Host-side code only: allocation, copies, the int** -> CSR flattening.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/macr_eval_compat.h"
#include "../../include/macr_hip.h"

namespace {
thread_local int g_status = 0;
thread_local char g_msg[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_msg, sizeof(g_msg), fmt, ap);
    va_end(ap);
    g_status = code;
    fprintf(stderr, "macr_eval_compat: %s\n", g_msg);
    return code;
}

struct DevBuf {                      // hipMalloc'd scratch freed on scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

#define HIP_TRY(expr, what)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(MACR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

int top_k_impl(const float *scores, int cols, int rows, int K, int *rankings) {
    if (!scores || !rankings || cols <= 0 || rows < 0 || K <= 0) return fail(MACR_E_INVALID, "c_top_k_array_index: bad argument");
    if (rows == 0) return 0;
    // rows are staged in slabs of at most 256 MiB of scores (a 4096 x 40981 batch of the reference is 671 MB)
    const size_t row_bytes = (size_t)cols * sizeof(float);
    int slab = (int)std::max<size_t>(1, (size_t(256) << 20) / row_bytes);
    slab = std::min(slab, rows);
    DevBuf d_scores, d_idx, d_ws;
    const size_t ws_bytes = macr_topk_scores_workspace_bytes(slab, K);      // top_k above 128: rounds of 128 positions
    HIP_TRY(d_scores.alloc((size_t)slab * row_bytes), "hipMalloc(scores)");
    HIP_TRY(d_idx.alloc((size_t)slab * K * sizeof(int)), "hipMalloc(rankings)");
    if (ws_bytes) HIP_TRY(d_ws.alloc(ws_bytes), "hipMalloc(workspace)");
    for (int r0 = 0; r0 < rows; r0 += slab) {
        const int n = std::min(slab, rows - r0);
        HIP_TRY(hipMemcpy(d_scores.p, scores + (size_t)r0 * cols, (size_t)n * row_bytes, hipMemcpyHostToDevice), "copy scores");
        const int rc = macr_topk_scores(d_scores.as<float>(), cols, n, K, d_idx.as<int32_t>(), nullptr, d_ws.p, ws_bytes, nullptr);
        if (rc != MACR_OK) return fail(rc, "c_top_k_array_index: %s", macr_last_error());
        HIP_TRY(hipMemcpy(rankings + (size_t)r0 * K, d_idx.p, (size_t)n * K * sizeof(int), hipMemcpyDeviceToHost), "copy rankings");
    }
    return 0;
}

int foldout_impl(int U, const int *rankings, int K, int **gts, const int *gt_num, float *results) {
    if (U < 0 || K <= 0 || !rankings || !gts || !gt_num || !results) return fail(MACR_E_INVALID, "evaluate_foldout: bad argument");
    if (U == 0) return 0;
    std::vector<int32_t> ptr((size_t)U + 1, 0), idx;
    for (int u = 0; u < U; ++u) {
        if (gt_num[u] < 0 || (gt_num[u] > 0 && !gts[u])) return fail(MACR_E_INVALID, "evaluate_foldout: ground truth of user %d", u);
        ptr[u + 1] = ptr[u] + gt_num[u];
    }
    idx.resize(std::max<size_t>(1, (size_t)ptr[U]));
    for (int u = 0; u < U; ++u) {                    // the kernel wants each user's ids ascending; the reference takes any order
        std::copy(gts[u], gts[u] + gt_num[u], idx.begin() + ptr[u]);
        std::sort(idx.begin() + ptr[u], idx.begin() + ptr[u + 1]);
    }
    DevBuf d_rank, d_ptr, d_idx, d_res;
    HIP_TRY(d_rank.alloc((size_t)U * K * sizeof(int)), "hipMalloc(rankings)");
    HIP_TRY(d_ptr.alloc(ptr.size() * sizeof(int32_t)), "hipMalloc(gt_ptr)");
    HIP_TRY(d_idx.alloc(idx.size() * sizeof(int32_t)), "hipMalloc(gt_idx)");
    HIP_TRY(d_res.alloc((size_t)U * 5 * K * sizeof(float)), "hipMalloc(results)");
    HIP_TRY(hipMemcpy(d_rank.p, rankings, (size_t)U * K * sizeof(int), hipMemcpyHostToDevice), "copy rankings");
    HIP_TRY(hipMemcpy(d_ptr.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice), "copy gt_ptr");
    HIP_TRY(hipMemcpy(d_idx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice), "copy gt_idx");
    const int rc = macr_metrics_foldout(U, K, d_rank.as<int32_t>(), d_ptr.as<int32_t>(), d_idx.as<int32_t>(), d_res.as<float>(), 0, nullptr);
    if (rc != MACR_OK) return fail(rc, "evaluate_foldout: %s", macr_last_error());
    HIP_TRY(hipMemcpy(results, d_res.p, (size_t)U * 5 * K * sizeof(float), hipMemcpyDeviceToHost), "copy results");
    return 0;
}
}  // namespace

extern "C" void c_top_k_array_index(float *scores_pt, int columns_num, int rows_num, int top_k, int thread_num, int *rankings_pt) {
    (void)thread_num;
    g_status = 0; g_msg[0] = 0;
    if (top_k_impl(scores_pt, columns_num, rows_num, top_k, rankings_pt) != 0 && rankings_pt && rows_num > 0 && top_k > 0)
        std::fill(rankings_pt, rankings_pt + (size_t)rows_num * top_k, -1);
}

extern "C" void evaluate_foldout(int users_num, int *rankings, int rank_len, int **ground_truths, int *ground_truths_num,
                                 int thread_num, float *results) {
    (void)thread_num;
    g_status = 0; g_msg[0] = 0;
    if (foldout_impl(users_num, rankings, rank_len, ground_truths, ground_truths_num, results) != 0 && results && users_num > 0 && rank_len > 0)
        std::fill(results, results + (size_t)users_num * 5 * rank_len, NAN);
}

extern "C" int macr_eval_compat_status(void) { return g_status; }
extern "C" const char *macr_eval_compat_error(void) { return g_msg; }
