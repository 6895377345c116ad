// Issue rate of the 32-bit integer multiply against the 24-bit one and a plain add on gfx950 (one wave per SIMD x 4
// waves per CU would do; here 256 CUs x 8 waves): 8 independent chains per lane, 4096 steps.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/imul_rate.hip -o /tmp/imul_rate ; run: /tmp/imul_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned a, unsigned b, int n) {
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i + a;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
            if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, unsigned* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 4096, blocks = 256 * 2;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, n);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * n * 8 / (256 CUs * 4 SIMDs)
    const double per_simd = (double)blocks * 4 * n * 8 / 1024.0;
    printf("%-14s %.3f ms  %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 2 * 256 * 4);
    run<2>("v_add_u32", d); run<1>("v_mul_u32_u24", d); run<4>("v_mad_u32_u24", d); run<0>("v_mul_lo_u32", d); run<3>("v_mul_hi_u32", d);
    return 0;
}
