// Micro-benchmark: does the bf16 MFMA SHAPE matter for conv_c8_kernel's inner loop?
// MI355X_MICROARCH.md "DVFS give-back" item 7: bare LDS-read + MFMA loops on random data deliver
// 1.12-1.14x with v_mfma_f32_16x16x32_bf16 over 32x32x16 at equal cycles (the chip holds a higher clock).
// Both kernels: 256 threads, ~70 KB LDS (2 workgroups per CU), wave tile 64 channels x 64 pixels, one
// barrier per 16-channel k-tile, operands re-read from LDS by ds_read_b128 for every tap (conv_c8's
// read pattern), no global traffic inside the loop.  Usage: mfma_shape [k-tiles] [workgroups]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WCH = 18 * 64;      // weight chunks per k-tile (tap, half, 64 channels)
constexpr int PW = 34, PCH = 10 * PW;   // patch: 10 rows x 34 columns per half
constexpr int LDS_CH = 2 * WCH + 2 * 1024;   // same footprint as conv_c8 RECT TN=2

__global__ __launch_bounds__(256, 2) void k32(const uint4* __restrict__ src, float* __restrict__ out, int nkt) {
    __shared__ __attribute__((aligned(16))) uint4 smem[LDS_CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int i = tid; i < LDS_CH; i += 256) smem[i] = src[(blockIdx.x * 37 + i) % 8192];
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int bpos[2] = {(wave * 2) * PW + l31, (wave * 2 + 1) * PW + l31};
    for (int kt = 0; kt < nkt; ++kt) {
        const int wb = kt & 1;
        const uint4* Ws = smem + wb * WCH;
        const uint4* Ps = smem + 2 * WCH + wb * 1024;
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            uint4 a[2], b[2];
            for (int i = 0; i < 2; ++i) a[i] = Ws[(tap * 2 + lh) * 64 + i * 32 + l31];
            for (int j = 0; j < 2; ++j) b[j] = Ps[lh * PCH + bpos[j] + ky * PW + kx];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                         __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

__global__ __launch_bounds__(256, 2) void k16(const uint4* __restrict__ src, float* __restrict__ out, int nkt) {
    __shared__ __attribute__((aligned(16))) uint4 smem[LDS_CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    for (int i = tid; i < LDS_CH; i += 256) smem[i] = src[(blockIdx.x * 37 + i) % 8192];
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    // wave's 64 pixels: 2 rows x 32 columns = 4 blocks of 16
    int bpos[4];
    for (int j = 0; j < 4; ++j) bpos[j] = (wave * 2 + (j >> 1)) * PW + (j & 1) * 16 + l15;
    const int h = g & 1, tsel = g >> 1;
    for (int kt = 0; kt < nkt; ++kt) {
        const int wb = kt & 1;
        const uint4* Ws = smem + wb * WCH;
        const uint4* Ps = smem + 2 * WCH + wb * 1024;
        __builtin_amdgcn_s_barrier();
        // 9 taps x 2 halves = 18 half-units; 4 per MFMA -> 4.5 steps per k-tile: taps (2m, 2m+1) for
        // m = 0..3, and tap 8 of two consecutive k-tiles together (here: every other k-tile runs a 5th step)
        const int nst = (kt & 1) ? 5 : 4;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            if (m < nst) {
                const int tap = m < 4 ? 2 * m + tsel : 8;
                const int ky = tap / 3, kx = tap - 3 * ky;
                uint4 a[4], b[4];
                for (int i = 0; i < 4; ++i) a[i] = Ws[(tap * 2 + h) * 64 + i * 16 + l15];
                for (int j = 0; j < 4; ++j) b[j] = Ps[h * PCH + bpos[j] + ky * PW + kx];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                             __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char** argv) {
    const int nkt = argc > 1 ? atoi(argv[1]) : 64;
    const int wgs = argc > 2 ? atoi(argv[2]) : 4096;
    std::vector<uint16_t> h(8192 * 8);
    srand(1);
    for (auto& v : h) {   // random bf16 in [-1, 1): sign, exponent 120..126, random mantissa
        v = (uint16_t)(((rand() & 1) << 15) | ((120 + rand() % 7) << 7) | (rand() & 127));
    }
    uint4* d; float* o;
    hipMalloc(&d, 8192 * 16); hipMalloc(&o, (size_t)wgs * 256 * 4);
    hipMemcpy(d, h.data(), 8192 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * 64 * 256 * 144 * (double)nkt * wgs;   // 64 ch x 256 px x (16 ch x 9 taps) per k-tile
    for (int round = 0; round < 4; ++round) {
        for (int which = 0; which < 2; ++which) {
            // ~1 s of back-to-back launches per arm so the clock settles
            float ms = 0; int n = 0;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                for (int q = 0; q < 8; ++q) {
                    if (which == 0) hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, d, o, nkt);
                    else hipLaunchKernelGGL(k16, dim3(wgs), dim3(256), 0, 0, d, o, nkt);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (rep >= 4) { ms += t; n += 8; }
            }
            printf("round %d %s: %.4f ms/launch  %.0f TFLOP/s\n", round, which ? "16x16x32" : "32x32x16", ms / n,
                   flop / (ms / n * 1e-3) / 1e12);
        }
    }
    return 0;
}
