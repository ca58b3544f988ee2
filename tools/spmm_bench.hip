// Kernel-development bench of the LightGCN propagation alone: builds in seconds (only spmm_kernels.hip), makes a
// synthetic bipartite graph with the law of macr_amd/synth.py (lognormal list lengths, Zipf items), runs
// macr_lgcn_propagate and reports per-launch HIP-event times + the error against a double-precision CPU product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics [-DMACR_ABL_...] tools/spmm_bench.hip -o tools/spmm_bench
//   tools/spmm_bench [n_users n_items mean_len d layers reps zipf]
#include "../macr_amd/csrc/capi_common.hip"
#include "../macr_amd/csrc/spmm_kernels.hip"

#include <cmath>
#include <map>
#include <random>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int n_u = argc > 1 ? atoi(argv[1]) : 31668, n_i = argc > 2 ? atoi(argv[2]) : 38048;
    const double mean = argc > 3 ? atof(argv[3]) : 1371000.0 / 31668;
    const int d = argc > 4 ? atoi(argv[4]) : 64, L = argc > 5 ? atoi(argv[5]) : 2, reps = argc > 6 ? atoi(argv[6]) : 30;
    const bool zipf = argc > 7 ? atoi(argv[7]) != 0 : true;
    const int N = n_u + n_i;
    std::mt19937_64 rng(9);
    std::lognormal_distribution<double> ln(std::log(std::max(mean, 1.0)) - 0.405, 0.9);
    std::vector<double> cdf(n_i);
    { double s = 0; for (int k = 0; k < n_i; ++k) { s += 1.0 / (k + 1); cdf[k] = s; } for (double &c : cdf) c /= s; }
    std::uniform_real_distribution<double> un(0.0, 1.0);
    std::vector<std::vector<int>> ul(n_u), il(n_i);
    size_t nnz_half = 0;
    for (int u = 0; u < n_u; ++u) {
        int n = (int)std::lround(ln(rng));
        n = std::min(std::max(n, 1), std::max(1, n_i / 2));
        std::vector<int> dr((size_t)(n * 1.5) + 4);
        for (int &x : dr) x = zipf ? (int)std::min<size_t>(std::lower_bound(cdf.begin(), cdf.end(), un(rng)) - cdf.begin(), n_i - 1) : (int)(un(rng) * n_i) % n_i;
        std::sort(dr.begin(), dr.end());
        dr.erase(std::unique(dr.begin(), dr.end()), dr.end());
        if ((int)dr.size() > n) dr.resize(n);
        ul[u] = dr; nnz_half += dr.size();
        for (int it : dr) il[it].push_back(u);
    }
    std::vector<int32_t> rowptr(N + 1, 0), col; std::vector<float> val;
    col.reserve(2 * nnz_half); val.reserve(2 * nnz_half);
    std::vector<double> dinv(N);
    for (int r = 0; r < N; ++r) { const size_t g = r < n_u ? ul[r].size() : il[r - n_u].size(); dinv[r] = g ? 1.0 / std::sqrt((double)g) : 0.0; }
    for (int r = 0; r < N; ++r) {
        if (r < n_u) for (int it : ul[r]) { col.push_back(n_u + it); val.push_back((float)(dinv[r] * dinv[n_u + it])); }
        else for (int u : il[r - n_u]) { col.push_back(u); val.push_back((float)(dinv[r] * dinv[u])); }
        rowptr[r + 1] = (int32_t)col.size();
    }
    const size_t nnz = col.size();
    std::vector<float> T((size_t)N * d);
    { std::uniform_real_distribution<float> ux(-1.f, 1.f); const float lim = std::sqrt(6.f / (N + d)); for (float &x : T) x = ux(rng) * lim; }
    if (getenv("BENCH_T_INDEX")) for (int r = 0; r < N; ++r) for (int k = 0; k < d; ++k) T[(size_t)r * d + k] = (float)r;

    const size_t pb = macr_spmm_plan_bytes(N, rowptr.data(), col.data(), val.data());
    std::vector<char> plan(pb);
    if (macr_spmm_plan_build(N, rowptr.data(), col.data(), val.data(), plan.data(), pb) != MACR_OK) { fprintf(stderr, "plan: %s\n", macr_last_error()); return 1; }
    const size_t wf = macr_lgcn_work_floats(N, d, plan.data());
    int32_t *d_rowptr, *d_col; float *d_val, *d_T, *d_E, *d_work; void *d_plan;
    CK(hipMalloc(&d_rowptr, (N + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4 + 256)); CK(hipMalloc(&d_val, nnz * 4 + 256));
    CK(hipMalloc(&d_T, T.size() * 4)); CK(hipMalloc(&d_E, T.size() * 4)); CK(hipMalloc(&d_work, wf * 4)); CK(hipMalloc(&d_plan, pb));
    CK(hipMemcpy(d_rowptr, rowptr.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_T, T.data(), T.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_plan, plan.data(), pb, hipMemcpyHostToDevice));
    CK(hipMemset(d_work, 0, wf * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto run = [&]() {
        if (macr_lgcn_propagate(N, d, L, d_rowptr, d_col, d_val, d_plan, plan.data(), d_T, d_E, d_work, st) != MACR_OK) {
            fprintf(stderr, "propagate: %s\n", macr_last_error()); exit(1);
        }
    };
    for (int k = 0; k < 5; ++k) run();
    CK(hipStreamSynchronize(st));
    std::map<std::string, std::pair<int, double>> agg;
    std::vector<std::string> order;
    for (int k = 0; k < reps; ++k) {
        macr_timing_begin(st);
        run();
        char names[64 * 32]; float ms[64];
        const int n = macr_timing_end(64, names, ms);
        for (int q = 0; q < n; ++q) {
            std::string nm = std::string(names + q * 32) + "#" + std::to_string(q);
            if (!agg.count(nm)) order.push_back(nm);
            agg[nm].first++; agg[nm].second += ms[q];
        }
    }
    // wall time of back-to-back propagations (no events in between)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st)); for (int k = 0; k < reps; ++k) run(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float wall = 0; CK(hipEventElapsedTime(&wall, e0, e1));
    // check
    std::vector<float> E(T.size());
    CK(hipMemcpy(E.data(), d_E, E.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> X(T.begin(), T.end()), S(T.begin(), T.end()), Y(T.size());
    for (int l = 0; l < L; ++l) {
        for (int r = 0; r < N; ++r)
            for (int k = 0; k < d; ++k) {
                double a = 0; for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) a += (double)val[e] * X[(size_t)col[e] * d + k];
                Y[(size_t)r * d + k] = a;
            }
        for (size_t q = 0; q < S.size(); ++q) S[q] += Y[q];
        X.swap(Y);
    }
    double err = 0; for (size_t q = 0; q < S.size(); ++q) err = std::max(err, std::fabs(S[q] / (L + 1) - (double)E[q]));
    if (err > 1e-5) {
        int bad = 0, shown = 0;
        for (int r = 0; r < N; ++r) {
            double e = 0; for (int k = 0; k < d; ++k) e = std::max(e, std::fabs(S[(size_t)r * d + k] / (L + 1) - (double)E[(size_t)r * d + k]));
            if (e > 1e-6) { ++bad; if (shown++ < 12) fprintf(stderr, "bad row %d (deg %d) err %.3g got %.5g want %.5g\n", r, rowptr[r + 1] - rowptr[r], e, E[(size_t)r * d], S[(size_t)r * d] / (L + 1)); }
        }
        fprintf(stderr, "%d bad rows of %d\n", bad, N);
        if (L == 1) for (int r = 0; r < 6; ++r) {
            const double got = 2.0 * E[(size_t)r * d] - T[(size_t)r * d];
            fprintf(stderr, "row %d got acc %.6g; prefix sums:", r, got);
            double a = 0;
            for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) { a += (double)val[e] * T[(size_t)col[e] * d]; fprintf(stderr, " %.6g", a); }
            fprintf(stderr, "\n   terms:");
            for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) fprintf(stderr, " %.6g(c%d w%.4g)", (double)val[e] * T[(size_t)col[e] * d], col[e], val[e]);
            fprintf(stderr, "\n");
        }
    }
#ifdef MACR_SPMM_TRACE
    {   // phase stamps of the LAST launch (the last dense layer): per class of row length, the mean cycles of each phase of a wave's
        // life, and how many waves were resident over the launch (100 MHz real-time stamps)
        const PlanHeader *phh = reinterpret_cast<const PlanHeader *>(plan.data());
        std::vector<unsigned long long> tr((size_t)(1 << 17) * 8);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(macr::g_spmm_trace), tr.size() * 8));
        const int nw = std::min(phh->n_single + phh->n_rec[0] + phh->n_rec[1] + phh->n_rec[2] + phh->n_rec[3], 1 << 17);   // (bundle waves have ids below n_items too)
        struct Acc { double n = 0, desc = 0, first = 0, rest = 0, store = 0, total = 0, entries = 0; };
        std::map<int, Acc> cls;
        unsigned long long t_min = ~0ull, t_max = 0;
        for (int w = 0; w < nw; ++w) { const auto *t = &tr[(size_t)w * 8]; if (t[0] && t[6]) { t_min = std::min(t_min, t[0]); t_max = std::max(t_max, t[6]); } }
        std::vector<int> resident((size_t)(t_max - t_min) + 2, 0);
        for (int w = 0; w < nw; ++w) {
            const auto *t = &tr[(size_t)w * 8];
            if (!t[0] || !t[6]) continue;
            const int len = (int)(t[7] >> 32), piece = (int)(t[7] & 1), bun = (int)((t[7] >> 1) & 7);
            const int c = bun ? 1000 * bun : piece ? -1 : len <= 8 ? 8 : len <= 32 ? 32 : len <= 64 ? 64 : len <= 128 ? 128 : 512;
            Acc &a = cls[c];
            a.n += 1; a.entries += len; a.desc += (double)(t[2] - t[1]);
            const bool multi = t[3] > t[2];
            a.first += multi ? (double)(t[3] - t[2]) : (double)(t[4] - t[2]);
            a.rest += multi ? (double)(t[4] - t[3]) : 0.0;
            a.store += (double)(t[5] - t[4]); a.total += (double)(t[5] - t[1]);
            for (unsigned long long x = t[0]; x <= t[6]; ++x) resident[(size_t)(x - t_min)]++;
        }
        fprintf(stderr, "trace of the last launch: %d waves, span %.2f us (100 MHz stamps)\n", nw, (t_max - t_min) / 100.0);
        for (auto &kv : cls) {
            const Acc &a = kv.second;
            fprintf(stderr, "  rows <= %4d%s: %6.0f waves, %5.1f entries; cycles: descriptor %6.0f | first window (index + gathers) %6.0f | further windows %7.0f | store %5.0f | life %7.0f\n",
                    kv.first < 0 ? 512 : kv.first, kv.first < 0 ? " (hub pieces)" : "", a.n, a.entries / a.n, a.desc / a.n, a.first / a.n, a.rest / a.n, a.store / a.n, a.total / a.n);
        }
        fprintf(stderr, "  resident waves per 1 us:");
        for (size_t x = 0; x + 100 <= resident.size(); x += 100) { long sum = 0; for (int q = 0; q < 100; ++q) sum += resident[x + q]; fprintf(stderr, " %ld", sum / 100); }
        fprintf(stderr, "\n");
    }
#endif
    const PlanHeader *ph = reinterpret_cast<const PlanHeader *>(plan.data());
    if (err > 1e-5 && ph->reserved > 0) {                        // walk the stream on the host: builder or kernel?
        const StreamHeader *shh = reinterpret_cast<const StreamHeader *>(reinterpret_cast<const int32_t *>(plan.data()) + ph->reserved);
        const StreamView v = view_stream(plan.data(), ph->reserved, *shh);
        std::vector<double> Xh(T.begin(), T.end()), Sh(T.begin(), T.end()), Yh(T.size(), 0.0), acc(d);
        std::vector<double> parts((size_t)shh->n_slots * d, 0.0);
        for (int w = 0; w < shh->n_chunks; ++w) {
            const int4 cd = v.chunk_desc[w];
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int e = cd.x * 32; e < cd.y * 32; ++e) {
                const int2 q = v.pcw[e];
                float wv; memcpy(&wv, &q.y, 4);
                if (q.x >= 0) { for (int k = 0; k < d; ++k) acc[k] += (double)wv * Xh[(size_t)q.x * d + k]; }
                else if (q.y == 1) { for (int k = 0; k < d; ++k) parts[(size_t)cd.w * d + k] = acc[k]; }
                else { const int r = q.x & 0x7fffffff; for (int k = 0; k < d; ++k) Yh[(size_t)r * d + k] = acc[k]; std::fill(acc.begin(), acc.end(), 0.0); }
            }
        }
        for (int k2 = 0; k2 < shh->n_split; ++k2)
            for (int g = v.split_group0[k2]; g < v.split_group0[k2 + 1]; ++g)
                for (int sl = v.group_slot0[g]; sl < v.group_slot0[g + 1]; ++sl)
                    for (int k = 0; k < d; ++k) Yh[(size_t)v.split_row[k2] * d + k] += parts[(size_t)sl * d + k];
        // compare with one reference layer
        double e1 = 0;
        for (int r = 0; r < N; ++r) for (int k = 0; k < d; ++k) {
            double a = 0; for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) a += (double)val[e] * (double)T[(size_t)col[e] * d + k];
            e1 = std::max(e1, std::fabs(a - Yh[(size_t)r * d + k]));
        }
        fprintf(stderr, "host walk of the stream, one layer: max err %.3g\n", e1);
    }
    int n_chunks = 0, s_slots = 0, s_entries = 0;
    if (ph->reserved > 0) {
        const StreamHeader *sh = reinterpret_cast<const StreamHeader *>(reinterpret_cast<const int32_t *>(plan.data()) + ph->reserved);
        n_chunks = sh->n_chunks; s_slots = sh->n_slots; s_entries = sh->n_entries;
    }
    printf("{\"N\": %d, \"nnz\": %zu, \"d\": %d, \"layers\": %d, \"plan_items\": %d, \"plan_slots\": %d, \"stream_chunks\": %d, \"stream_slots\": %d, \"stream_entries\": %d, \"wall_us_per_propagate\": %.2f, \"max_err\": %.3g, \"kernels_us\": {",
           N, nnz, d, L, ph->n_items, ph->n_slots, n_chunks, s_slots, s_entries, 1e3 * wall / reps, err);
    for (size_t q = 0; q < order.size(); ++q) printf("%s\"%s\": %.2f", q ? ", " : "", order[q].c_str(), 1e3 * agg[order[q]].second / agg[order[q]].first);
    printf("}}\n");
    return err < 1e-5 ? 0 : 2;
}
