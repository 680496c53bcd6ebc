// Probe (GPU box): which fp32 expression is v_dot2c_f32_f16 / v_dot2c_f32_bf16 on gfx950?
// Result on MI355X (1 M random samples): none of the simple candidates is exact - the closest, the exact
// sum rounded once, matches 97.3 % (f16) / 99.2 % (bf16).  The low-precision residue is therefore checked
// against a tolerance (tests/test_gpu_parity.py), not against a bit-level twin.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dot2 tools/probes/dot2_semantics.hip ; run /tmp/dot2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__global__ void k16(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a[i]), __builtin_bit_cast(h2, b[i]), c[i], false);
}
__global__ void kbf(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a[i]), __builtin_bit_cast(b2, b[i]), c[i], false);
}
static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    const int n = 1 << 20;
    std::mt19937 rng(7);
    std::vector<uint32_t> a(n), b(n); std::vector<float> c(n), o(n);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < n; ++i) {
            auto half = [&]() { float v = std::ldexp((float)(rng() & 0xFFFF) / 65536.0f, (int)(rng() % 6) - 3) * ((rng() & 1) ? 1 : -1);
                                 if (mode == 0) { _Float16 x = (_Float16)v; uint16_t u; memcpy(&u, &x, 2); return u; }
                                 uint32_t w; memcpy(&w, &v, 4); return (uint16_t)(w >> 16); };
            a[i] = half() | ((uint32_t)half() << 16); b[i] = half() | ((uint32_t)half() << 16);
            c[i] = std::ldexp((float)(rng() & 0xFFFFFF) / 16777216.0f, (int)(rng() % 8) - 2) * ((rng() & 1) ? 1 : -1);
        }
        uint32_t *da, *db; float *dc, *dout;
        (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dc, n * 4); (void)hipMalloc(&dout, n * 4);
        (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
        if (mode == 0) hipLaunchKernelGGL(k16, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
        else hipLaunchKernelGGL(kbf, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
        (void)hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long m[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            auto cv = [&](uint16_t h) { return mode == 0 ? h2f(h) : bf2f(h); };
            const float ax = cv(a[i] & 0xFFFF), ay = cv(a[i] >> 16), bx = cv(b[i] & 0xFFFF), by = cv(b[i] >> 16);
            const float p0 = ax * bx, p1 = ay * by;  // exact for f16; rounded for bf16? (8x8 bits: exact too)
            const float cand[6] = {
                std::fmaf(ay, by, std::fmaf(ax, bx, c[i])),            // 0: fma chain x then y
                std::fmaf(ax, bx, std::fmaf(ay, by, c[i])),            // 1: fma chain y then x
                (p0 + p1) + c[i],                                      // 2: products summed first
                (float)((double)p0 + (double)p1 + (double)c[i]),       // 3: single rounding of the exact sum
                (c[i] + p0) + p1,                                      // 4
                (c[i] + p1) + p0 };                                    // 5
            for (int t = 0; t < 6; ++t) m[t] += memcmp(&cand[t], &o[i], 4) == 0;
        }
        printf("%s: n=%d  fma(x),fma(y)=%ld  fma(y),fma(x)=%ld  (p0+p1)+c=%ld  exact-sum-rounded-once=%ld  (c+p0)+p1=%ld  (c+p1)+p0=%ld\n",
               mode == 0 ? "f16" : "bf16", n, m[0], m[1], m[2], m[3], m[4], m[5]);
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
    }
    return 0;
}
