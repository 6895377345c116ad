// Micro-benchmark: conv_c8_kernel's inner loop is short of LDS read bandwidth (2 A + 2 B
// ds_read_b128 per 4 MFMAs on the 256-pixel forms).  With 32-pixel tile rows the B operand of tap (ky, kx + 1)
// is the B operand of (ky, kx) one lane over: does reading B once per tap ROW and shifting it by DPP
// (v_mov_b32_dpp wave_shl:1, lanes 31 / 63 patched from a two-lane read of columns 32, 33) beat three reads?
//   ref    the kernel's read pattern: per tap 2 A + 2 B reads, 4 MFMAs
//   shift  per tap row 2 B reads + 4 two-lane reads, 16 DPP moves + 16 selects; per tap 2 A reads, 4 MFMAs
//   nob    upper bound: B read once per tap row and reused unshifted (wrong values, no shift cost)
// All: 256 threads, the LDS footprint of conv_c8 RECT TN = 2 (2 workgroups per CU), wave tile 64 ch x 64 px,
// one barrier per k-tile, no global traffic in the loop.  Usage: lds_shift [k-tiles] [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int WCH = 18 * 64;
constexpr int PW = 34, PCH = 10 * PW;
constexpr int LDS_CH = 2 * WCH + 2 * 1024;

__device__ __forceinline__ u32x4 shl1(u32x4 v, u32x4 tail, bool last) {
    // lane l <- lane l + 1 (wave_shl:1 = 0x130); lanes 31 / 63 <- the tail column
    u32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned s = (unsigned)__builtin_amdgcn_update_dpp((int)tail[q], (int)v[q], 0x130, 0xf, 0xf, false);
        r[q] = last ? tail[q] : s;
    }
    return r;
}

template <int MODE>   // 0 ref, 1 shift, 2 nob
__global__ __launch_bounds__(256, 2) void kern(const uint4* __restrict__ src, float* __restrict__ out, int nkt) {
    __shared__ __attribute__((aligned(16))) uint4 smem[LDS_CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < LDS_CH; i += 256) smem[i] = src[(blockIdx.x * 37 + i) % 8192];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int bpos[2] = {(wave * 2) * PW + l31, (wave * 2 + 1) * PW + l31};
    const bool last = l31 == 31;
    for (int kt = 0; kt < nkt; ++kt) {
        const int wb = kt & 1;
        const uint4* Ws = smem + wb * WCH;
        const uint4* Ps = smem + 2 * WCH + wb * 1024;
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            u32x4 b[3][2];
            if (MODE == 0) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    for (int j = 0; j < 2; ++j) b[kx][j] = __builtin_bit_cast(u32x4, Ps[lh * PCH + bpos[j] + ky * PW + kx]);
            } else {
                u32x4 t1[2], t2[2];
                for (int j = 0; j < 2; ++j) {
                    b[0][j] = __builtin_bit_cast(u32x4, Ps[lh * PCH + bpos[j] + ky * PW]);
                    t1[j] = b[0][j]; t2[j] = b[0][j];
                }
                if (MODE == 1) {
                    if (last) {
                        for (int j = 0; j < 2; ++j) {
                            t1[j] = __builtin_bit_cast(u32x4, Ps[lh * PCH + bpos[j] + ky * PW + 1]);
                            t2[j] = __builtin_bit_cast(u32x4, Ps[lh * PCH + bpos[j] + ky * PW + 2]);
                        }
                    }
                    for (int j = 0; j < 2; ++j) {
                        b[1][j] = shl1(b[0][j], t1[j], last);
                        b[2][j] = shl1(b[1][j], t2[j], last);
                    }
                } else {
                    for (int j = 0; j < 2; ++j) { b[1][j] = b[0][j]; b[2][j] = b[0][j]; }
                }
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tap = ky * 3 + kx;
                uint4 a[2];
                for (int i = 0; i < 2; ++i) a[i] = Ws[(tap * 2 + lh) * 64 + i * 32 + l31];
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                             __builtin_bit_cast(bf16x8, b[kx][j]),
                                                                             acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const int nkt = argc > 1 ? atoi(argv[1]) : 64;
    const int wgs = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<uint16_t> h(8192 * 8);
    srand(1);
    for (auto& v : h) v = (uint16_t)(((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127));
    uint4* d; float* o;
    hipMalloc(&d, 8192 * 16); hipMalloc(&o, (size_t)wgs * 256 * 4);
    hipMemcpy(d, h.data(), 8192 * 16, hipMemcpyHostToDevice);
    // the shifted operands must be the values the three reads deliver
    std::vector<float> r0((size_t)wgs * 256), r1((size_t)wgs * 256);
    hipLaunchKernelGGL(kern<0>, dim3(wgs), dim3(256), 0, 0, d, o, 4);
    hipMemcpy(r0.data(), o, r0.size() * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(kern<1>, dim3(wgs), dim3(256), 0, 0, d, o, 4);
    hipMemcpy(r1.data(), o, r1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("shift vs ref: %zu of %zu sums differ\n", bad, r0.size());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * 64 * 256 * 144 * (double)nkt * wgs;
    const char* names[3] = {"ref  ", "shift", "nob  "};
    for (int round = 0; round < 3; ++round)
        for (int which = 0; which < 3; ++which) {
            float ms = 0; int n = 0;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                for (int q = 0; q < 8; ++q) {
                    if (which == 0) hipLaunchKernelGGL(kern<0>, dim3(wgs), dim3(256), 0, 0, d, o, nkt);
                    else if (which == 1) hipLaunchKernelGGL(kern<1>, dim3(wgs), dim3(256), 0, 0, d, o, nkt);
                    else hipLaunchKernelGGL(kern<2>, dim3(wgs), dim3(256), 0, 0, d, o, nkt);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep >= 4) { ms += t; n += 8; }
            }
            printf("round %d %s: %.4f ms/launch  %.0f TFLOP/s\n", round, names[which], ms / n, flop / (ms / n * 1e-3) / 1e12);
        }
    return 0;
}
