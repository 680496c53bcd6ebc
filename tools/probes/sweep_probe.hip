// Lab probe (not shipped): the sweep kernel alone on a Bernoulli pattern.  Two builds: plain (the time per launch
// is the shipped kernel's) and -DBSMR_SWEEP_STAMPS (in-kernel clock stamps and parts of the consumer loop left
// out by a mask; the stamps cost hundreds of cycles per barrier, so that build's time is not the kernel's).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DBSMR_SWEEP_STAMPS] -Iinclude -Ibsmr-sddmm_amd/csrc \
//         -o tools/probes/sweep_probe[_stamps] tools/probes/sweep_probe.hip
//   tools/probes/sweep_probe M N density variant strip_blocks [iters]
// variant: one of the names in the table below (K, panels per wave, consumer waves, fp32 or 16-bit operands).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#ifndef BSMR_SWEEP_STAMPS
#define BSMR_SWEEP_LAB
#endif
#include "sweep_kernels.hpp"

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

using bsmr::SweepItem;
#ifdef BSMR_SWEEP_STAMPS
typedef void (*Kernel)(const void*, const void*, const uint32_t*, const SweepItem*, const uint32_t*, const uint32_t*,
                       const uint32_t*, float*, uint32_t, bsmr::Batch, uint64_t*, uint32_t);
#define LAB_ARGS(stamps) , stamps, skip
#else
typedef void (*Kernel)(const void*, const void*, const uint32_t*, const SweepItem*, const uint32_t*, const uint32_t*,
                       const uint32_t*, float*, uint32_t, bsmr::Batch, uint32_t);
#define LAB_ARGS(stamps) , skip
#define BSMR_SWEEP_LAB_BUILD
#endif
struct Variant {
    const char* name;
    int KS, PW, W, NL, perCu, NBB;
    bool src32;
    int mode;
    Kernel kernel;
};
#define V(name, KS, PW, MODE, SRC, W, NL, PC, NBB) {name, KS, PW, W, NL, PC, NBB, SRC, MODE, bsmr::denseSweep<KS, PW, MODE, SRC, W, NL, PC, NBB>}
static const Variant kVariants[] = {
    V("k128_f32_w4p4_n4b1", 4, 4, 0, true, 4, 4, 1, 1), V("k128_f32_w4p4_n4b2", 4, 4, 0, true, 4, 4, 1, 2),
    V("k128_f32_w4p4_n8b1", 4, 4, 0, true, 4, 8, 1, 1), V("k128_f32_w4p4_n8b2", 4, 4, 0, true, 4, 8, 1, 2),
    V("k128_f32_w4p4_n8b4", 4, 4, 0, true, 4, 8, 1, 4), V("k128_f32_w8p2_n8b2", 4, 2, 0, true, 8, 8, 1, 2),
    V("k128_f32_w8p2_n8b4", 4, 2, 0, true, 8, 8, 1, 4), V("k128_h_w4p4_n8b2", 4, 4, 0, false, 4, 8, 1, 2),
    V("k128_h_w4p4_n4b2", 4, 4, 0, false, 4, 4, 1, 2), V("k128_h_w4p4_n4b4", 4, 4, 0, false, 4, 4, 1, 4),
    V("k512_b_w4p2_n4b1", 16, 2, 1, false, 4, 4, 1, 1), V("k512_b_w4p2_n4b2", 16, 2, 1, false, 4, 4, 1, 2),
    V("k512_b_w8p1_n4b2", 16, 1, 1, false, 8, 4, 1, 2), V("k512_b_w8p1_n8b2", 16, 1, 1, false, 8, 8, 1, 2),
    V("k512_b_w8p2_n4b2", 16, 2, 1, false, 8, 4, 1, 2),
};



static uint16_t toF16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t toBf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float fromF16(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static float fromBf16(uint16_t u) { uint32_t x = (uint32_t)u << 16; float f; memcpy(&f, &x, 4); return f; }

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s M N density variant strip_blocks [iters [skip mask]]\n", argv[0]);
        return 2;
    }
    const uint32_t M = atoi(argv[1]), N = atoi(argv[2]);
    const double density = atof(argv[3]);
    const Variant* v = nullptr;
    for (const Variant& x : kVariants)
        if (std::string(x.name) == argv[4]) v = &x;
    if (!v) {
        fprintf(stderr, "unknown variant\n");
        return 2;
    }
    const uint32_t SB = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 50;
    const uint32_t skip = argc > 7 ? (uint32_t)strtoul(argv[7], nullptr, 0) : 0u;   // lab: parts of the consumer loop left out
    const uint32_t K = 32u * v->KS;

    // pattern: Bernoulli(density), rows in natural order
    std::mt19937 rng(4);
    std::vector<uint32_t> ro(M + 1, 0), ci;
    {
        std::bernoulli_distribution coin(density);
        for (uint32_t i = 0; i < M; ++i) {
            for (uint32_t j = 0; j < N; ++j)
                if (coin(rng)) ci.push_back(j);
            ro[i + 1] = (uint32_t)ci.size();
        }
    }
    const uint32_t nnz = (uint32_t)ci.size();
    bsmr::HostDense hd;
    hd.M = M; hd.N = N; hd.nnz = nnz; hd.numPanels = (M + 15) / 16;
    hd.panelRows.resize((size_t)hd.numPanels * 16);
    for (size_t i = 0; i < hd.panelRows.size(); ++i) hd.panelRows[i] = i < M ? (uint32_t)i : 0u;
    hd.offsets.assign(hd.numPanels + 1, 0);
    for (uint32_t p = 0; p < hd.numPanels; ++p) {
        struct E { uint32_t col, row, idx; };
        std::vector<E> es;
        for (uint32_t r = 0; r < 16 && p * 16 + r < M; ++r)
            for (uint32_t e = ro[p * 16 + r]; e < ro[p * 16 + r + 1]; ++e) es.push_back({ci[e], r, e});
        std::sort(es.begin(), es.end(), [](const E& a, const E& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
        for (const E& e : es) { hd.col.push_back(e.col); hd.row.push_back((uint8_t)e.row); hd.idx.push_back(e.idx); }
        hd.offsets[p + 1] = hd.col.size();
    }
    bsmr::SweepFormatHost f;
    int st = bsmr::packSweep(hd, v->W, v->PW, SB, f);
    if (st != 0) {
        fprintf(stderr, "packSweep: %d\n", st);
        return 1;
    }
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    std::uniform_real_distribution<float> u(0.f, 2.f);
    for (float& x : A) x = u(rng);
    for (float& x : B) x = u(rng);
    std::vector<uint16_t> A16(A.size()), B16(B.size());
    for (size_t i = 0; i < A.size(); ++i) A16[i] = v->mode ? toBf16(A[i]) : toF16(A[i]);
    for (size_t i = 0; i < B.size(); ++i) B16[i] = v->mode ? toBf16(B[i]) : toF16(B[i]);

    void *dA, *dB;
    uint32_t *dRows, *dStarts, *dRowStart, *dWords;
    SweepItem* dItems;
    float* dP;
    uint64_t* dStamps;
    const size_t esz = v->src32 ? 4 : 2;
    CHECK(hipMalloc(&dA, A.size() * esz));
    CHECK(hipMalloc(&dB, B.size() * esz));
    CHECK(hipMemcpy(dA, v->src32 ? (void*)A.data() : (void*)A16.data(), A.size() * esz, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, v->src32 ? (void*)B.data() : (void*)B16.data(), B.size() * esz, hipMemcpyHostToDevice));
#define UP(dst, vec) CHECK(hipMalloc((void**)&dst, vec.size() * sizeof(vec[0]))); CHECK(hipMemcpy(dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice))
    UP(dRows, f.panelRows); UP(dItems, f.items); UP(dStarts, f.starts); UP(dRowStart, f.rowStart); UP(dWords, f.words);
    CHECK(hipMalloc((void**)&dP, (size_t)nnz * 4));
    CHECK(hipMemset(dP, 0xFF, (size_t)nnz * 4));
    const uint32_t numItems = (uint32_t)f.items.size();
    CHECK(hipMalloc((void**)&dStamps, (size_t)numItems * 32 * 8));
    CHECK(hipMemset(dStamps, 0, (size_t)numItems * 32 * 8));
    const size_t lds = bsmr::sweepLdsBytes(v->KS, v->PW, v->src32, v->W, v->perCu, v->NBB);
    CHECK(hipFuncSetAttribute((const void*)v->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const bsmr::Batch batch{0, 0, 0, 1};
    const dim3 grid(numItems), block((v->W + v->NL) * 64);
    printf("%s: M %u N %u nnz %u K %u | items %u (groups %u x strips %u) blocks/strip %u | lds %zu slots %u | max step entries %u, item words %u\n", v->name, M,
           N, nnz, K, numItems, f.numGroups, f.numStrips, SB, lds, bsmr::sweepSlots(v->KS, v->src32, v->W, v->PW, v->perCu, v->NBB), f.maxStepEntries, f.maxItemWords);
    if (f.maxItemWords + 64 > bsmr::kSweepItemWords) {
        fprintf(stderr, "an item's entry words do not fit the LDS region\n");
        return 1;
    }
    hipLaunchKernelGGL(v->kernel, grid, block, lds, 0, dA, dB, dRows, dItems, dStarts, dRowStart, dWords, dP, N, batch LAB_ARGS(dStamps));
    CHECK(hipDeviceSynchronize());
    // check against the rounded-operand fp64 sum
    std::vector<float> P(nnz);
    CHECK(hipMemcpy(P.data(), dP, (size_t)nnz * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    uint32_t bad = 0;
    for (uint32_t i = 0; i < M; ++i)
        for (uint32_t e = ro[i]; e < ro[i + 1]; ++e) {
            double s = 0;
            for (uint32_t k = 0; k < K; ++k) {
                const float a = v->mode ? fromBf16(A16[(size_t)i * K + k]) : fromF16(A16[(size_t)i * K + k]);
                const float b = v->mode ? fromBf16(B16[(size_t)ci[e] * K + k]) : fromF16(B16[(size_t)ci[e] * K + k]);
                s += (double)a * b;
            }
            const double rel = std::fabs(P[e] - s) / std::max(1e-3, std::fabs(s));
            if (!(rel < 1e-4)) ++bad;
            if (rel == rel) worst = std::max(worst, rel);
        }
    printf("check: %u of %u entries off (worst relative error %.3g)\n", bad, nnz, worst);
    // timing
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(v->kernel, grid, block, lds, 0, dA, dB, dRows, dItems, dStarts, dRowStart, dWords, dP, N, batch LAB_ARGS((uint64_t*)nullptr));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(v->kernel, grid, block, lds, 0, dA, dB, dRows, dItems, dStarts, dRowStart, dWords, dP, N, batch LAB_ARGS((uint64_t*)nullptr));
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("time: %.2f us per launch\n", ms * 1000.f / iters);
#ifndef BSMR_SWEEP_STAMPS
    return 0;
#endif
    // stamps of a launch in the middle of a busy stream
    hipLaunchKernelGGL(v->kernel, grid, block, lds, 0, dA, dB, dRows, dItems, dStarts, dRowStart, dWords, dP, N, batch LAB_ARGS((uint64_t*)nullptr));
    hipLaunchKernelGGL(v->kernel, grid, block, lds, 0, dA, dB, dRows, dItems, dStarts, dRowStart, dWords, dP, N, batch LAB_ARGS(dStamps));
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)numItems * 32);
    CHECK(hipMemcpy(h.data(), dStamps, h.size() * 8, hipMemcpyDeviceToHost));
    uint64_t first = ~0ull, last = 0;
    for (uint32_t i = 0; i < numItems; ++i) {
        first = std::min(first, std::min(h[(size_t)i * 32], h[(size_t)i * 32 + 16]));
        last = std::max(last, std::max(h[(size_t)i * 32 + 3], h[(size_t)i * 32 + 21]));
    }
    auto med = [&](auto fn) {
        std::vector<double> x(numItems);
        for (uint32_t i = 0; i < numItems; ++i) x[i] = fn(&h[(size_t)i * 32]);
        std::sort(x.begin(), x.end());
        printf("median %9.0f  p10 %9.0f  p90 %9.0f\n", x[x.size() / 2], x[x.size() / 10], x[x.size() * 9 / 10]);
    };
    printf("clock ticks (s_memtime), %u items, first start to last end: %llu\n", numItems, (unsigned long long)(last - first));
    printf("  consumer start offset      "); med([&](const uint64_t* q) { return (double)(q[0] - first); });
    printf("  consumer prologue loads    "); med([&](const uint64_t* q) { return (double)(q[1] - q[0]); });
    printf("  consumer panel phase       "); med([&](const uint64_t* q) { return (double)(q[2] - q[1]); });
    printf("  consumer block phase       "); med([&](const uint64_t* q) { return (double)(q[3] - q[2]); });
    printf("    of it at the barrier     "); med([&](const uint64_t* q) { return (double)q[4]; });
    printf("  consumer lifetime          "); med([&](const uint64_t* q) { return (double)(q[3] - q[0]); });
    printf("  loader prologue loads      "); med([&](const uint64_t* q) { return (double)(q[17] - q[16]); });
        printf("  loader first image landed  "); med([&](const uint64_t* q) { return (double)(q[19] - q[18]); });
    printf("  loader panels done         "); med([&](const uint64_t* q) { return (double)(q[20] - q[18]); });
    printf("  loader end                 "); med([&](const uint64_t* q) { return (double)(q[21] - q[18]); });
    printf("    loop: waiting on vmcnt   "); med([&](const uint64_t* q) { return (double)q[22]; });
    printf("    loop: at the barrier     "); med([&](const uint64_t* q) { return (double)q[23]; });
    return 0;
}
