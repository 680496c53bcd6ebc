// Lab probe (not shipped): how fast can ONE CU take in L2-resident bytes?  Every workgroup streams `bytesPerWg`
// contiguous bytes of a small table (so that the XCD's L2 serves it) with NW waves, each keeping DEPTH 1-KiB
// LDS-DMA pieces (or 16-byte register loads) in flight.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/stream_probe tools/probes/stream_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                              \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
            return 1;                                                         \
        }                                                                     \
    } while (0)

template <uint32_t N>
__device__ __forceinline__ void waitVm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA: wave w of the workgroup moves pieces w, w + NW, ... of the workgroup's region; DEPTH pieces in flight per wave
template <int NW, int DEPTH, bool SWIZZLE>
__global__ void __launch_bounds__(NW * 64) streamDma(const uint8_t* __restrict__ table, uint32_t tableBytes, uint32_t bytesPerWg,
                                                      uint32_t wgStride, uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    // blocks b and b + 8 share an XCD: give the workgroups of one XCD neighbouring regions
    const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const uint32_t wg = xcd * (gridDim.x >> 3) + idx;
    const uint32_t base = (uint32_t)(((uint64_t)wg * wgStride) % (tableBytes - bytesPerWg));
    const uint32_t pieces = bytesPerWg / 1024u;
    uint8_t* mine = lds + wave * (DEPTH * 1024u);
    const uint32_t lanePiece = SWIZZLE ? ((lane & ~15u) | ((lane & 15u) ^ ((lane >> 5) & 15u) ^ 5u)) : lane;
    uint32_t issued = 0;
    for (uint32_t p = wave; p < pieces; p += NW) {
        const uint8_t* src = table + base + p * 1024u + lanePiece * 16u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(mine + (issued % DEPTH) * 1024u), 16, 0, 0);
        ++issued;
        if (issued >= DEPTH) waitVm<DEPTH - 1>();
    }
    waitVm<0>();
    if (sink && lane == 0 && wave == 0) sink[blockIdx.x] = *reinterpret_cast<uint32_t*>(mine);
}

// register loads: DEPTH x 16 bytes per lane in flight
template <int NW, int DEPTH>
__global__ void __launch_bounds__(NW * 64) streamReg(const uint8_t* __restrict__ table, uint32_t tableBytes, uint32_t bytesPerWg,
                                                      uint32_t wgStride, uint32_t* __restrict__ sink) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const uint32_t wg = xcd * (gridDim.x >> 3) + idx;
    const uint32_t base = (uint32_t)(((uint64_t)wg * wgStride) % (tableBytes - bytesPerWg));
    const uint32_t pieces = bytesPerWg / 1024u;
    uint4 acc = {0, 0, 0, 0};
    for (uint32_t p = wave * DEPTH; p + DEPTH <= pieces; p += NW * DEPTH) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const uint4*>(table + base + (p + d) * 1024u + lane * 16u);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
    if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc.x;
}

// what a workgroup barrier costs: NW waves meet `rounds` times, with `work` dependent VALU operations in between
template <int NW>
__global__ void __launch_bounds__(NW * 64) barrierLoop(uint32_t rounds, uint32_t work, uint32_t* __restrict__ sink) {
    uint32_t x = threadIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        for (uint32_t w = 0; w < work; ++w) x = x * 1664525u + 1013904223u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (sink && x == 0x12345678u) sink[blockIdx.x] = x;
}

template <typename F>
static int timeIt(const char* name, F launch, double bytes, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / iters;
    printf("  %-34s %8.2f us  %7.2f TB/s  %6.1f GB/s per CU\n", name, us, bytes / us * 1e-6, bytes / us * 1e-3 / 256.0);
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t tableBytes = argc > 1 ? atoi(argv[1]) << 10 : 6400u << 10;    // KiB
    const uint32_t bytesPerWg = argc > 2 ? atoi(argv[2]) << 10 : 288u << 10;     // KiB per workgroup
    const uint32_t wgStride = argc > 3 ? atoi(argv[3]) << 10 : 24u << 10;         // KiB between neighbouring workgroups' regions
    const uint32_t wgs = argc > 4 ? atoi(argv[4]) : 256;
    uint8_t* table;
    uint32_t* sink;
    CHECK(hipMalloc((void**)&table, tableBytes));
    CHECK(hipMemset(table, 1, tableBytes));
    CHECK(hipMalloc((void**)&sink, 4096 * 4));
    const double bytes = (double)wgs * bytesPerWg;
    printf("table %u KiB, %u KiB per workgroup, stride %u KiB, %u workgroups: %.1f MB per launch\n", tableBytes >> 10, bytesPerWg >> 10,
           wgStride >> 10, wgs, bytes * 1e-6);
#define DMA(NW, DEPTH, SWZ)                                                                                                  \
    {                                                                                                                        \
        auto k = streamDma<NW, DEPTH, SWZ>;                                                                                  \
        CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
        char name[64];                                                                                                       \
        snprintf(name, 64, "dma  waves %2d depth %2d%s", NW, DEPTH, SWZ ? " swizzled" : "");                                 \
        if (timeIt(name, [&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(NW * 64), NW * DEPTH * 1024, 0, table, tableBytes, bytesPerWg, wgStride, sink); }, bytes, 20)) return 1; \
    }
#define REG(NW, DEPTH)                                                                                                       \
    {                                                                                                                        \
        auto k = streamReg<NW, DEPTH>;                                                                                       \
        char name[64];                                                                                                       \
        snprintf(name, 64, "reg  waves %2d depth %2d", NW, DEPTH);                                                           \
        if (timeIt(name, [&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(NW * 64), 0, 0, table, tableBytes, bytesPerWg, wgStride, sink); }, bytes, 20)) return 1; \
    }
#define BAR(NW, ROUNDS, WORK, LDS)                                                                                             \
    {                                                                                                                        \
        auto k = barrierLoop<NW>;                                                                                            \
        CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
        char name[64];                                                                                                       \
        snprintf(name, 64, "barrier waves %2d rounds %3d work %3d lds %3dK", NW, ROUNDS, WORK, LDS);                            \
        if (timeIt(name, [&] { hipLaunchKernelGGL(k, dim3(wgs), dim3(NW * 64), LDS * 1024, 0, ROUNDS, WORK, sink); }, 0.0, 20)) return 1; \
    }
    BAR(4, 0, 0, 0) BAR(4, 40, 0, 0) BAR(4, 400, 0, 0) BAR(8, 40, 0, 0) BAR(8, 400, 0, 0) BAR(12, 0, 0, 0) BAR(12, 40, 0, 0) BAR(12, 400, 0, 0)
    BAR(12, 0, 0, 150) BAR(12, 40, 0, 150) BAR(12, 40, 50, 150) BAR(12, 400, 0, 150)
    DMA(1, 8, false) DMA(1, 32, false) DMA(2, 16, false) DMA(4, 4, false) DMA(4, 8, false) DMA(4, 16, false) DMA(4, 32, false)
    DMA(8, 4, false) DMA(8, 8, false) DMA(8, 16, false) DMA(16, 4, false) DMA(16, 8, false) DMA(4, 16, true) DMA(8, 8, true)
    REG(4, 4) REG(4, 8) REG(8, 4) REG(8, 8) REG(16, 4) REG(16, 8)
    return 0;
}
