// Lab probe (not shipped): the output-stationary GEMM kernel (csrc/gemm_kernels.hpp) alone on a Bernoulli pattern,
// rows in natural order.  Every written value is checked against a CPU sum over the rounded operands; -DBSMR_GEMM_LAB
// builds take a mask that leaves parts of the kernel out (timing only: results are then wrong and not checked).
//   tools/probes/build.sh
//   tools/probes/gemm_probe M N density variant [iters [skip mask]]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "gemm_kernels.hpp"

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

using bsmr::GemmItem;
#ifdef BSMR_GEMM_LAB
typedef void (*Kernel)(const void*, const void*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, const GemmItem*, const uint32_t*,
                       const uint32_t*, const uint32_t*, float*, uint32_t, uint32_t, uint32_t, uint32_t, bsmr::Batch, uint32_t, unsigned long long*);
#define LAB_ARGS , skip, stampsArg
#else
typedef void (*Kernel)(const void*, const void*, uint32_t, uint32_t, const uint32_t*, const uint32_t*, const GemmItem*, const uint32_t*,
                       const uint32_t*, const uint32_t*, float*, uint32_t, uint32_t, uint32_t, uint32_t, bsmr::Batch);
#define LAB_ARGS
#endif
struct Variant {
    const char* name;
    int KT, PM, NB, mode;
    bool src32;
    Kernel kernel;
    bool shared = false;   // denseGemmCvt: one rounding per element through a second LDS image
};
#define V(name, KT, PM, NB, MODE) {name, KT, PM, NB, MODE, false, bsmr::denseGemm<KT, PM, NB, MODE, false>}
#define V32(name, KT, PM, NB, MODE) {name, KT, PM, NB, MODE, true, bsmr::denseGemm<KT, PM, NB, MODE, true>}
#define VC(name, KT, PM, NB, MODE) {name, KT, PM, NB, MODE, true, reinterpret_cast<Kernel>(bsmr::denseGemmCvt<KT, PM, NB, MODE>), true}
static const Variant kVariants[] = {
    V("k512_b_256x256", 8, 16, 16, 1), V("k512_b_128x256", 8, 8, 16, 1), V("k512_b_256x128", 8, 16, 8, 1),
    V("k512_b_256x320", 8, 16, 20, 1), V("k512_b_256x192", 8, 16, 12, 1), V("k512_b_128x320", 8, 8, 20, 1),
    V("k256_h_256x256", 4, 16, 16, 0), V("k128_h_256x256", 2, 16, 16, 0), V("k128_h_128x256", 2, 8, 16, 0),
    V("k128_h_256x128", 2, 16, 8, 0), V("k128_h_256x320", 2, 16, 20, 0), V("k128_h_256x192", 2, 16, 12, 0),
    V("k64_h_256x256", 1, 16, 16, 0),
    // fp32 operands, rounded in the kernel (KT = K / 32)
    V32("k128_f32_256x256", 4, 16, 16, 0), V32("k128_f32_256x320", 4, 16, 20, 0), V32("k128_f32_128x256", 4, 8, 16, 0),
    V32("k128_f32_128x320", 4, 8, 20, 0), V32("k64_f32_256x256", 2, 16, 16, 0), V32("k32_f32_256x256", 1, 16, 16, 0),
    V32("k128_f32b_256x320", 4, 16, 20, 1),
    // fp32 operands, one shared rounding per element (denseGemmCvt)
    VC("k128_cvt_256x256", 4, 16, 16, 0), VC("k128_cvt_256x320", 4, 16, 20, 0), VC("k128_cvt_128x256", 4, 8, 16, 0),
    VC("k128_cvt_128x320", 4, 8, 20, 0), VC("k64_cvt_256x320", 2, 16, 20, 0), VC("k128_cvtb_256x320", 4, 16, 20, 1),
    VC("k32_cvt_256x320", 1, 16, 20, 0), VC("k512_cvtb_256x256", 16, 16, 16, 1), VC("k256_cvt_256x256", 8, 16, 16, 0),
    VC("k512_cvtb_256x320", 16, 16, 20, 1), VC("k512_cvt_256x320", 16, 16, 20, 0),
};

static uint16_t toF16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t toBf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float fromF16(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static float fromBf16(uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return f; }

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s M N density variant [iters [skip mask]]\n", argv[0]);
        return 2;
    }
    const uint32_t M = atoi(argv[1]), N = atoi(argv[2]);
    const double density = atof(argv[3]);
    const Variant* v = nullptr;
    for (const Variant& x : kVariants)
        if (std::string(x.name) == argv[4]) v = &x;
    if (!v) {
        fprintf(stderr, "variants:");
        for (const Variant& x : kVariants) fprintf(stderr, " %s", x.name);
        fprintf(stderr, "\n");
        return 2;
    }
    const int iters = argc > 5 ? atoi(argv[5]) : 200;
    const uint32_t skip = argc > 6 ? (uint32_t)strtoul(argv[6], nullptr, 0) : 0u;
    (void)skip;
    const uint32_t K = (v->src32 ? 32u : 64u) * v->KT;

    // pattern: i.i.d. Bernoulli(density), CSR with sorted rows; panels in natural row order
    std::mt19937 rng(4);
    std::bernoulli_distribution coin(density);
    std::vector<uint32_t> ro(M + 1, 0), ci;
    const bool zipf = getenv("GEMM_ZIPF") != nullptr;   // column j kept with probability min(1, c / (j + 1)): hot first columns (nips-like)
    double zc = 0;
    if (zipf) {   // c such that the expected row length is density * N
        double lo = 0, hi = N;
        for (int it = 0; it < 60; ++it) {
            zc = 0.5 * (lo + hi);
            double sum = 0;
            for (uint32_t j = 0; j < N; ++j) sum += std::min(1.0, zc / (j + 1));
            (sum < density * N ? lo : hi) = zc;
        }
    }
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (uint32_t i = 0; i < M; ++i) {
        for (uint32_t j = 0; j < N; ++j)
            if (zipf ? uni(rng) < std::min(1.0, zc / (j + 1)) : coin(rng)) ci.push_back(j);
        ro[i + 1] = (uint32_t)ci.size();
    }
    const uint32_t nnz = (uint32_t)ci.size();
    bsmr::HostDense hd;
    hd.M = M; hd.N = N; hd.nnz = nnz; hd.numPanels = (M + 15) / 16;
    hd.panelRows.assign((size_t)hd.numPanels * 16, 0);
    for (uint32_t i = 0; i < hd.numPanels * 16; ++i) hd.panelRows[i] = i < M ? i : 0;
    hd.offsets.assign(hd.numPanels + 1, 0);
    for (uint32_t p = 0; p < hd.numPanels; ++p) {   // (column, row) order inside a panel
        std::vector<std::pair<uint64_t, uint32_t>> list;
        for (uint32_t rr = 0; rr < 16 && p * 16 + rr < M; ++rr)
            for (uint32_t e = ro[p * 16 + rr]; e < ro[p * 16 + rr + 1]; ++e) list.push_back({((uint64_t)ci[e] << 8) | rr, e});
        std::sort(list.begin(), list.end());
        for (auto& x : list) {
            hd.col.push_back((uint32_t)(x.first >> 8));
            hd.row.push_back((uint8_t)(x.first & 255));
            hd.idx.push_back(x.second);
        }
        hd.offsets[p + 1] = hd.col.size();
    }
    bsmr::GemmFormatHost f;
    const int st = bsmr::packGemm(hd, v->PM, v->NB, f, !getenv("GEMM_NATURAL"));   // GEMM_NATURAL=1: columns in natural order
    if (st != BSMR_OK) {
        fprintf(stderr, "packGemm: %d\n", st);
        return 1;
    }

    // operands U[0,2), rounded to the 16-bit type on the host
    std::vector<uint16_t> A16((size_t)M * K), B16((size_t)N * K);
    std::vector<float> Af(A16.size()), Bf(B16.size());
    std::mt19937 gen(5489);
    auto draw = [&]() { return 2.0f * (float)(gen() >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> A32(v->src32 ? A16.size() : 0), B32(v->src32 ? B16.size() : 0);   // what the fp32 kernel reads
    for (size_t i = 0; i < A16.size(); ++i) { const float x = draw(); if (v->src32) A32[i] = x; A16[i] = v->mode ? toBf16(x) : toF16(x); Af[i] = v->mode ? fromBf16(A16[i]) : fromF16(A16[i]); }
    for (size_t i = 0; i < B16.size(); ++i) { const float x = draw(); if (v->src32) B32[i] = x; B16[i] = v->mode ? toBf16(x) : toF16(x); Bf[i] = v->mode ? fromBf16(B16[i]) : fromF16(B16[i]); }

    uint8_t *dA, *dB;
    const size_t esz = v->src32 ? 4 : 2;
    uint32_t *dRows, *dColOf, *dRowStart, *dLists, *dWords;
    GemmItem* dItems;
    float* dP;
    CHECK(hipMalloc(&dA, A16.size() * esz)); CHECK(hipMalloc(&dB, B16.size() * esz));
    CHECK(hipMalloc(&dRows, f.panelRows.size() * 4)); CHECK(hipMalloc(&dItems, f.items.size() * sizeof(GemmItem)));
    CHECK(hipMalloc(&dColOf, f.colOf.size() * 4));
    CHECK(hipMemcpy(dColOf, f.colOf.data(), f.colOf.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dRowStart, f.rowStart.size() * 4)); CHECK(hipMalloc(&dLists, f.lists.size() * 4));
    CHECK(hipMalloc(&dWords, f.words.size() * 4)); CHECK(hipMalloc(&dP, (size_t)nnz * 4));
    CHECK(hipMemcpy(dA, v->src32 ? (const void*)A32.data() : (const void*)A16.data(), A16.size() * esz, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, v->src32 ? (const void*)B32.data() : (const void*)B16.data(), B16.size() * esz, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRows, f.panelRows.data(), f.panelRows.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dItems, f.items.data(), f.items.size() * sizeof(GemmItem), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dRowStart, f.rowStart.data(), f.rowStart.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dLists, f.lists.data(), f.lists.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dWords, f.words.data(), f.words.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dP, 0xFF, (size_t)nnz * 4));

    const size_t lds = v->shared ? bsmr::gemmCvtLdsBytes(v->PM, v->NB) : bsmr::gemmLdsBytes(v->PM, v->NB);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(v->kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const bsmr::Batch batch{0, 0, 0, 1};
    const uint32_t full = f.fullGrid && !getenv("GEMM_ITEMS") ? 1u : 0u;   // GEMM_ITEMS=1: take the places from the item records
    unsigned long long* stampsArg = nullptr;   // lab builds: where one launch leaves its clock readings
    (void)stampsArg;
    auto launch = [&]() {
        hipLaunchKernelGGL(v->kernel, dim3((uint32_t)f.items.size()), dim3(512), lds, nullptr, dA, dB, (uint32_t)(A16.size() * esz),
                           (uint32_t)(B16.size() * esz), dRows, dColOf, dItems, dRowStart, dLists, dWords, dP, N, f.numGroups, f.numStrips, full, batch LAB_ARGS);
    };
    launch();
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::vector<float> P(nnz);
    CHECK(hipMemcpy(P.data(), dP, (size_t)nnz * 4, hipMemcpyDeviceToHost));
    // check: every entry written; a sample against the fp64 sum of the rounded operands
    uint64_t unwritten = 0, bad = 0;
    double maxRel = 0;
    if (!skip) {
        for (uint32_t e = 0; e < nnz; ++e) unwritten += std::isnan(P[e]);
        const uint32_t stride = std::max<uint32_t>(1, nnz / 200000);
        for (uint32_t i = 0; i < M; ++i)
            for (uint32_t e = ro[i]; e < ro[i + 1]; ++e) {
                if (e % stride) continue;
                double s = 0;
                const float* a = &Af[(size_t)i * K];
                const float* b = &Bf[(size_t)ci[e] * K];
                for (uint32_t k = 0; k < K; ++k) s += (double)a[k] * b[k];
                const double rel = std::fabs(P[e] - s) / std::max(std::fabs(s), 1e-3);
                maxRel = std::max(maxRel, rel);
                if (!(rel < 1e-4)) {
                    if (bad < 5) fprintf(stderr, "entry %u (row %u col %u): got %g want %g\n", e, i, ci[e], P[e], s);
                    ++bad;
                }
            }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        for (int i = 0; i < 10; ++i) launch();
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / iters);
        sum += ms / iters;
    }
#ifdef BSMR_GEMM_LAB
    {   // one launch with clock readings (100 MHz): where the waves are when, relative to the first wave's entry
        const size_t count = f.items.size() * 8 * 8;
        unsigned long long* dStamps;
        CHECK(hipMalloc(&dStamps, count * 8));
        CHECK(hipMemset(dStamps, 0, count * 8));
        for (int i = 0; i < 10; ++i) launch();
        stampsArg = dStamps;
        launch();
        stampsArg = nullptr;
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(count);
        CHECK(hipMemcpy(st.data(), dStamps, count * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < count / 8; ++w) t0 = std::min(t0, st[w * 8]);
        static const char* kNames[8] = {"entry", "first stage requested", "own share landed", "first barrier passed", "K loop left", "pass 0 done", "pass 1 done", "pass 2 done"};
        printf("  clock readings over %zu waves, us after the first wave's entry (min / mean / max):\n", count / 8);
        for (int i = 0; i < 8; ++i) {
            double lo = 1e30, hi = 0, sum = 0;
            size_t seen = 0;
            for (size_t w = 0; w < count / 8; ++w) {
                if (!st[w * 8 + i]) continue;
                const double us = (double)(st[w * 8 + i] - t0) * 0.01;
                lo = std::min(lo, us); hi = std::max(hi, us); sum += us; ++seen;
            }
            if (seen) printf("    %-22s %6.2f %6.2f %6.2f\n", kNames[i], lo, sum / seen, hi);
        }
        // the workgroups' last readings, sorted: how the launch drains
        std::vector<double> ends;
        for (size_t g = 0; g < f.items.size(); ++g) {
            unsigned long long e = 0;
            for (size_t w = 0; w < 8; ++w)
                for (int i = 4; i < 8; ++i) e = std::max(e, st[(g * 8 + w) * 8 + i]);
            ends.push_back((double)(e - t0) * 0.01);
        }
        std::sort(ends.begin(), ends.end());
        printf("    workgroups done: 10 %% %.2f, 50 %% %.2f, 90 %% %.2f, 99 %% %.2f, last %.2f\n", ends[ends.size() / 10], ends[ends.size() / 2],
               ends[ends.size() * 9 / 10], ends[ends.size() * 99 / 100], ends.back());
    }
#endif
    const double flops = 2.0 * (double)f.numTiles * 256.0 * K;
    printf("%s M=%u N=%u nnz=%u items=%zu tiles=%llu words=%zu lds=%zu skip=0x%x: %.2f us best, %.2f us mean; executed %.1f TFLOP/s, useful %.1f; "
           "unwritten %llu, bad %llu, max rel %.2e\n",
           v->name, M, N, nnz, f.items.size(), (unsigned long long)f.numTiles, f.words.size(), lds, skip, best * 1e3, sum / 5 * 1e3,
           flops / (best * 1e-3) / 1e12, 2.0 * nnz * K / (best * 1e-3) / 1e12, (unsigned long long)unwritten, (unsigned long long)bad, maxRel);
    return unwritten || bad ? 3 : 0;
}
