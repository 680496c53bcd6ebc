// Self-test of tools/check_isa.py (`make check-isa`): a kernel that makes round 2's mistake on purpose.  The register an
// inline-assembly load fills is carried around a loop and read (accumulated) at the top of the next iteration, before
// the wait that covers the load - the checker must flag it; `isaTrapFixed` waits first and must pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void isaTrap(const u32x4* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    u32x4 v = {0, 0, 0, 0};
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        sum += v[0] + v[3];                                   // reads what the PREVIOUS iteration's load may not have delivered yet
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(in + i * 64 + threadIdx.x) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[threadIdx.x] = sum + v[1];
}

__global__ void isaTrapFixed(const u32x4* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
    u32x4 v = {0, 0, 0, 0};
    uint32_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(in + i * 64 + threadIdx.x) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
        sum += v[0] + v[3];
    }
    out[threadIdx.x] = sum;
}
