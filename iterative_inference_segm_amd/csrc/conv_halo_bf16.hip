// Halo-tile direct 3x3 convolution on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32
// accumulation): the 16-bit MFMA kernel for the SHALLOW 3x3 layers of the hot path (few channels,
// 113^2..422^2 maps), which are HBM-bound once the matrix work is 16x cheaper than in fp32.
// Same tiling as conv_halo.hip: a workgroup owns a TH x 32 pixel tile of one image and BM output
// channels; per k-tile (16 input channels) it stages the (TH+2) x 34 input patch ONCE, here as bf16
// "k8 chunks" (the 8 channels one lane feeds to an MFMA are 16 contiguous bytes):
//     Ps[h][py][px] = bf16(x[c0 + 8h .. c0 + 8h + 7][iy0 + py][ix0 + px])      h = 0, 1
// built in registers (8 coalesced dword loads -> 4 v_cvt_pk_bf16_f32 -> 1 ds_write_b128; with
// IISEG_CONV_UNPOOL the DePool2D equality mask of layers/mylayers.py:88-115 is applied to the fp32
// values first: 3 loads + compare per patch element).  The MFMA B operand of tap (ky, kx) is one
// conflict-free ds_read_b128 at Ps[lane >> 5][y + ky][x + kx]; the A operand comes from the packed
// bf16 weights Wp16[kt][tap][h][co][8] streamed global -> LDS by 16-byte LDS-DMA.
// Activations stay fp32 NCHW in HBM; fusions as in conv_halo.hip (bias / skip-add with crop / ReLU /
// window / placement / channel slice, 2x2 max-pool in the epilogue).  Statistical parity only.
// Measured dead ends (round 2, 128 -> 128 channels on 119^2 windows, 0.57 ms): 16-row tiles at two
// workgroups per CU 6 % slower; a one-time stagger of co-resident workgroups no change.  PMC on that
// launch: matrix pipe 20 % busy, 53 % of wave time parked on s_waitcnt / barriers, L2 hit 77 %, mean
// L2 latency 335 cycles -- a chain of short dependent phases per 16-channel k-tile, not a bandwidth
// limit; LDS operand reads one tap ahead of the MFMAs and wave-uniform scalar offsets (no waterfall
// loops around the patch loads) bought 4-13 %.
// Same Lasagne Conv2DLayer call sites (models/fcn8.py:34-45, models/fcn_down.py:102-104,
// models/fcn_up.py:83-86; FC-DenseNet's BN_ReLU_Conv convs, models/FCDenseNet.py:90).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include "iiseg.h"
#include "common.h"
#include "conv_common.h"

using namespace iiseg;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access, 4-byte aligned
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RSRC_W3 = 0x00027000;
constexpr unsigned OOB = 0x80000000u;
constexpr int CPT = 16;                 // channels per k-tile = one MFMA k-step per tap

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, RSRC_W3);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// Wp16[kt][tap][h][co][j] = bf16(w[co][c = 16 kt + 8 h + j][tap]), zero beyond Cin / Cout
__global__ void halo_pack_bf16_kernel(const float* __restrict__ w, int64_t so, int64_t sc,
                                      __bf16* __restrict__ wp, int Cin, int Cout, int nkt, int Mpad) {
    const int64_t n = (int64_t)nkt * 9 * 2 * Mpad * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        int64_t r = i >> 3;
        const int co = (int)(r % Mpad); r /= Mpad;
        const int h = (int)(r & 1); r >>= 1;
        const int tap = (int)(r % 9);
        const int kt = (int)(r / 9);
        const int c = kt * CPT + h * 8 + j;
        const float v = (c < Cin && co < Cout) ? w[co * so + c * sc + tap] : 0.f;
        wp[i] = (__bf16)v;
    }
}

// MASKIN (with UNPOOL): DePool2D mask from bytes (ConvParams::mask_in) instead of pre == pooled
template <int BM, int TH, int WM, int WN, bool UNPOOL, bool MASKIN = false>
__global__ __launch_bounds__(256, (UNPOOL && !MASKIN) ? 2 : 3) void conv_halo_bf16_kernel(const ConvParams p, const int tiles_y,
                                                                const int tiles_x) {
    constexpr int TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int NCHK = 2 * PP;                  // patch chunks per k-tile
    constexpr int NE = (NCHK + 255) / 256;        // ... per thread
    // MASKIN: a thread stages one POOLED position (8 channels of `up` + their 8 mask bytes, 16 loads)
    // and writes the up to four patch chunks of its 2x2 block -- a third of the loads of the
    // per-pixel form, which fetched every `up` value and mask byte once per pixel of the block
    constexpr int QH = PH / 2 + 1, QW = PW / 2 + 1, QP = QH * QW;
    static_assert(!MASKIN || 2 * QP <= 256, "one pooled position per thread");
    constexpr int NL = MASKIN ? 1 : NE;           // (chunk, channel) load slots per thread / 8
    constexpr int WTM = BM / WM, TM = WTM / 32;
    constexpr int RW = TH / WN, TN = RW;          // output rows per wave = 32-pixel MFMA column tiles
    constexpr int WCH = 9 * 2 * BM;               // weight chunks per k-tile
    constexpr int WPT = (WCH + 255) / 256;
    static_assert(WM * WN == 4 && TH % WN == 0 && BM % (WM * 32) == 0, "tile config");

    // ONE LDS array (a second __shared__ object can make hipcc drain the DMA queue early), carved
    // into the weight ring Ws[2][WCH] ([tap][h][BM] chunks), the single patch buffer Ps[NCHK]
    // ([h][PH][PW] chunks; 47.7 KB per workgroup at BM = 64: three workgroups per CU) and, in the
    // epilogue, the output staging tile Cs[32][TH][32] floats over the same bytes.
    constexpr int CSCH = 32 * TH * 32 / 4;            // chunks of the staging tile (32 KB at TH = 8)
    constexpr int SMCH = (2 * WCH + NCHK) > CSCH ? (2 * WCH + NCHK) : CSCH;
    __shared__ __attribute__((aligned(16))) uint4 smem[SMCH];
    uint4 (*Ws)[WCH] = reinterpret_cast<uint4 (*)[WCH]>(smem);
    uint4 (*Ps)[NCHK] = reinterpret_cast<uint4 (*)[NCHK]>(smem + 2 * WCH);

    int pt, mt;
    tile_of_block(blockIdx.x, gridDim.x, p.n_ptiles, p.n_mtiles, pt, mt);
    const int m0 = mt * BM;
    const int tpi = tiles_y * tiles_x;
    const int b = pt / tpi;
    const int tr = pt - b * tpi;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int wy0 = ty * TH, wx0 = tx * TW;                        // tile origin, window coords
    const int iy0 = p.oy0 + wy0 - p.pad, ix0 = p.ox0 + wx0 - p.pad;  // patch origin, input coords

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int HW = p.H * p.W, hw2 = p.h2 * p.w2;
    const int C1 = p.C1, Ctot = p.C1 + p.C2;

    // ---- patch staging: chunk e = i*256 + tid -> (channel half h, patch y, patch x) -------------
    unsigned voff[NE], voff2[UNPOOL ? NE : 1];
    int bsel[UNPOOL ? NE : 1];
    int hbits = 0;                                // bit i: chunk i holds channels 8..15 of the k-tile
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = i * 256 + tid;
        const int h = e / PP, rr = e - h * PP;
        const int py = rr / PW, px = rr - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        bool ok = e < NCHK && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        hbits |= h << i;
        if (p.debug_nogather & 2) ok = false;
        // (the chunk's channel half goes into the PER-LANE offset: the scalar offset of a load is
        // then wave-uniform -- a lane-dependent one makes hipcc wrap every load in a waterfall loop)
        voff[i] = ok ? 4u * (unsigned)(iy * p.W + ix + 8 * h * HW) : OOB;
        if constexpr (UNPOOL) {
            // DePool2D (layers/mylayers.py:95-114): only the 2h x 2w region has pooling windows
            ok = ok && iy < 2 * p.h2 && ix < 2 * p.w2;
            voff2[i] = ok ? 4u * (unsigned)((iy >> 1) * p.w2 + (ix >> 1) + 8 * h * hw2) : OOB;
            bsel[i] = ((iy & 1) << 1) | (ix & 1);
        }
    }
    unsigned q_voff = OOB;
    int q_slot[4] = {-1, -1, -1, -1}, q_h = 0;
    if constexpr (MASKIN) {
        q_h = tid / QP;
        const int qr = tid - q_h * QP, qy = qr / QW, qx = qr - qy * QW;
        const int Y2 = (iy0 >> 1) + qy, X2 = (ix0 >> 1) + qx;       // (arithmetic shifts: floor)
        const bool qin = tid < 2 * QP;
        // DePool2D (layers/mylayers.py:95-114): outside the h2 x w2 pooled map there is no window
        // (padding, the odd trailing row / column): the chunks written there are zero
        if (qin && (unsigned)Y2 < (unsigned)p.h2 && (unsigned)X2 < (unsigned)p.w2 && !(p.debug_nogather & 2))
            q_voff = 4u * (unsigned)(Y2 * p.w2 + X2 + 8 * q_h * hw2);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const int py = 2 * Y2 + (sl >> 1) - iy0, px = 2 * X2 + (sl & 1) - ix0;
            if (qin && (unsigned)py < (unsigned)PH && (unsigned)px < (unsigned)PW)
                q_slot[sl] = (q_h * PH + py) * PW + px;
        }
    }
    const unsigned char* basem = MASKIN ? p.mask_in + (size_t)b * C1 * hw2 : nullptr;
    // one image per tile: descriptors start at image b of each source
    const float* base1 = (UNPOOL && !MASKIN) ? p.pre + (size_t)b * C1 * HW : p.x1 + (size_t)b * C1 * HW;
    const float* base2 = p.C2 > 0 ? p.x2 + (size_t)b * p.C2 * HW : base1;
    const int n1 = C1 * HW * 4, n2 = p.C2 > 0 ? p.C2 * HW * 4 : n1;
    const float* baseq = UNPOOL ? p.pooled + (size_t)b * C1 * hw2 : nullptr;
    const float* baseu = UNPOOL ? p.x1 + (size_t)b * C1 * hw2 : nullptr;
    const int nq = C1 * hw2 * 4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xv[NE][8];
    float xq[UNPOOL ? NL : 1][8], xu[UNPOOL ? NL : 1][8];
    const int nkt = p.Kpad;                        // bf16 plan: Kpad holds the number of k-tiles
    static_assert(WCH % 64 == 0, "a wave's DMA piece is whole");
    const __amdgpu_buffer_rsrc_t wrsrc = mk_rsrc(p.wp, (p.debug_nogather & 1) ? 0 : nkt * 9 * 2 * p.Mpad * 16);

    // channels [c0, c0+16) of the logical (concatenated) input; the source is tile-uniform
    // (C1 % 16 == 0 when there are two sources); channels beyond the layer's read as zero
    // Branch-free on purpose, also past the last k-tile (every channel out of range: the loads return
    // zeros without memory traffic).  HBF_LOAD_X_SLICE issues the loads n = 8 i + j of one slice
    // [N0, N1): the main loop spreads the next k-tile's loads over the taps, between the MFMAs.
#define HBF_LOAD_X(KT) HBF_LOAD_X_SLICE(KT, 0, NL * 8)
#define HBF_LOAD_X_SLICE(KT, N0, N1)                                                               \
    {                                                                                              \
        const int c0 = (KT) * CPT;                                                                 \
        const bool s1 = UNPOOL || c0 < C1;                                                         \
        const int crem = (s1 ? C1 : Ctot) - c0;                                                    \
        const int cb = s1 ? c0 : c0 - C1;                                                          \
        static_for<((N0) < NL * 8 ? (N0) : NL * 8), ((N1) < NL * 8 ? (N1) : NL * 8)>([&](auto N) __attribute__((always_inline)) { \
            constexpr int i = decltype(N)::value / 8, j = decltype(N)::value % 8;                  \
            {                                                                                      \
                const bool cok = ((MASKIN ? q_h : ((hbits >> i) & 1)) * 8 + j) < crem;             \
                const unsigned so = (unsigned)((cb + j) * HW) * 4u;                                \
                if constexpr (!MASKIN)                                                             \
                    xv[i][j] = buf_ld(mk_rsrc(s1 ? base1 : base2, s1 ? n1 : n2), cok ? voff[i] : OOB, so); \
                if constexpr (UNPOOL) {                                                            \
                    const unsigned so2 = (unsigned)((cb + j) * hw2) * 4u;                          \
                    const unsigned vo2 = cok ? (MASKIN ? q_voff : voff2[i]) : OOB;                 \
                    if constexpr (MASKIN)                                                          \
                        xq[i][j] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b8( \
                            mk_rsrc(basem, nq >> 2), (int)(vo2 == OOB ? OOB : vo2 >> 2), (int)(so2 >> 2), 0)); \
                    else                                                                           \
                        xq[i][j] = buf_ld(mk_rsrc(baseq, nq), vo2, so2);                           \
                    xu[i][j] = buf_ld(mk_rsrc(baseu, nq), vo2, so2);                               \
                }                                                                                  \
            }                                                                                      \
        });                                                                                        \
    }
#define HBF_STORE_X(BUF)                                                                           \
    if constexpr (MASKIN) {                                                                        \
        _Pragma("unroll") for (int sl = 0; sl < 4; ++sl) {                                         \
            /* bit (row & 1) * 2 + (col & 1) of the window's byte: pre == pooled */                \
            float v[8];                                                                            \
            _Pragma("unroll") for (int j = 0; j < 8; ++j)                                          \
                v[j] = ((__builtin_bit_cast(unsigned, xq[0][j]) >> sl) & 1u) ? xu[0][j] : 0.f;     \
            if (q_slot[sl] >= 0)                                                                   \
                Ps[BUF][q_slot[sl]] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),     \
                                                 pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));    \
        }                                                                                          \
    } else                                                                                         \
    static_for<0, NE>([&](auto I) __attribute__((always_inline)) {                                 \
        constexpr int i = decltype(I)::value;                                                      \
        float v[8];                                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
            /* padding / odd trailing row+col read 0 == 0 -> up, which is also 0 there */          \
            if constexpr (MASKIN)                                                                  \
                v[j] = ((__builtin_bit_cast(unsigned, xq[i][j]) >> bsel[i]) & 1u) ? xu[i][j] : 0.f; \
            else if constexpr (UNPOOL) v[j] = (xv[i][j] == xq[i][j]) ? xu[i][j] : 0.f;             \
            else v[j] = xv[i][j];                                                                  \
        }                                                                                          \
        if (NE * 256 == NCHK || i * 256 + tid < NCHK)                                              \
            Ps[BUF][i * 256 + tid] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]),      \
                                                pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));     \
    });
    // weights of k-tile KT, channels [m0, m0 + BM): rows (tap, h) of BM chunks each
    // (the last, partial piece first and alone behind its wave-uniform test; the full pieces are
    // unconditional so that they share a basic block with the patch loads and the MFMAs.  Past the
    // last k-tile the offsets are out of the descriptor's range: zeros land in the idle ring half.)
#define HBF_LOAD_W(KT, BUF)                                                                        \
    static_for<0, WPT>([&](auto J) __attribute__((always_inline)) {                                \
        constexpr int j = WPT - 1 - decltype(J)::value;                                            \
        const int f = j * 256 + tid;                                                               \
        const int row = f / BM, col = f % BM;                                                      \
        if ((j + 1) * 256 <= WCH || j * 256 + wave * 64 < WCH)                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(                                              \
                wrsrc, (__attribute__((address_space(3))) void*)(&Ws[BUF][0] + j * 256 + wave * 64), \
                16, (int)(16u * (unsigned)(((KT) * 18 + row) * p.Mpad + m0 + col)), 0, 0, 0);      \
    });

    HBF_LOAD_X(0)
    HBF_LOAD_W(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HBF_STORE_X(0)
    __syncthreads();

    const int lrow = wn * RW;   // first output row of this wave inside the tile
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nkt;
        // The next k-tile's loads are spread over the nine taps, a slice between the LDS reads and
        // the MFMAs of each (issued back to back in front of the MFMAs they took 2-4 thousand cycles
        // per k-tile, stamped in-kernel -- longer than the MFMAs themselves), the weight DMA last:
        // hipcc drains the DMA queue in front of any later LDS read, and after tap 8 there is none.
        // operands of tap t+1 are read from LDS while the MFMAs of tap t run (two register sets)
        uint4 a[2][TM], bq[2][TN];
        auto lds_operands = [&](auto TAP, auto SET) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value, set = decltype(SET)::value;
            constexpr int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[set][i] = Ws[buf][(tap * 2 + lh) * BM + wm * WTM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bq[set][j] = Ps[0][(lh * PH + lrow + j + ky) * PW + l31 + kx];
        };
        lds_operands(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 9>([&](auto TAP) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value;
            constexpr int NPT = (NL * 8 + 7) / 8;          // patch loads (i, j) per tap, taps 0..7
            if constexpr (tap + 1 < 9)
                lds_operands(std::integral_constant<int, tap + 1>{},
                             std::integral_constant<int, (tap + 1) & 1>{});
            if constexpr (tap < 8) HBF_LOAD_X_SLICE(kt + 1, tap * NPT, (tap + 1) * NPT)
            if constexpr (tap == 8) HBF_LOAD_W(kt + 1, buf ^ 1)
            if (!(p.debug_nogather & 4)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, a[tap & 1][i]),
                        __builtin_bit_cast(bf16x8, bq[tap & 1][j]), acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);             // keep that order tap by tap
        });
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                 // every wave has read the patch of k-tile kt
            HBF_STORE_X(0)
        }
        __syncthreads();
    }
#undef HBF_LOAD_X
#undef HBF_LOAD_X_SLICE
#undef HBF_STORE_X
#undef HBF_LOAD_W
    // The last k-tile still issued its (out-of-range: zeros) patch loads and the weight DMA into the
    // idle ring half, and nothing waited for them (`more` was false).  The staging tile Cs below
    // overlays those LDS bytes, and a barrier does not drain ANOTHER wave's in-flight LDS-DMA: every
    // wave retires its own vector-memory queue before the first epilogue barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: bias, skip add (center-cropped), ReLU, NCHW store ---------------------------
    // The MFMA C/D layout gives a lane one pixel of 16 different channels: stored as it stands,
    // every store instruction writes two 128-byte pieces of two planes with 4 bytes per lane, and
    // measured (ablation on one device) those stores were 54 % of this kernel's time.  So the tile
    // takes one trip through LDS, 32 channels at a time: Cs[co][row][x] <- acc (conflict-free: lanes
    // are consecutive x), then every thread moves 16-byte pieces (4 consecutive pixels of a row):
    // 8 lanes cover a 32-pixel row of one plane, skip-add values come in as 16-byte loads, 4x fewer
    // store instructions, whole 128-byte lines wherever the row start allows.
    // C/D layout of the 32x32 MFMA: column = lane & 31 (pixel x), row = (r&3) + 8*(r>>2) + 4*lh
    static_assert(WM == 1, "the staging tile holds all rows of one 32-channel block");
    static_assert(TH == 8, "the store / pool passes below map a wave to one channel of the staged tile");
    // All epilogue traffic goes through buffer descriptors of image b with 32-bit offsets computed
    // ONCE per thread (stamped in-kernel, the first version of this epilogue spent 35-40 % of a
    // workgroup's cycles here, mostly on per-piece 64-bit address arithmetic and bounds branches):
    // out-of-range pieces get the out-of-bounds offset instead of a branch.
    float* Cs = reinterpret_cast<float*>(smem);
    const int OPL = p.out_H * p.out_W, APL = p.AH * p.AW, PPL = p.pool_H * p.pool_W;
    if ((p.debug_nogather & 8) && acc[0][0][0] != 12345.f) return;
    const bool pooling = p.pool != nullptr;
    const __amdgpu_buffer_rsrc_t r_bias = mk_rsrc(p.bias, p.bias ? p.Cout * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_out =
        mk_rsrc(p.out ? p.out + (size_t)b * p.out_ctot * OPL : nullptr, p.out ? p.out_ctot * OPL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_add =
        mk_rsrc(p.add ? p.add + (size_t)b * p.Cout * APL : nullptr, p.add ? p.Cout * APL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_pool =
        mk_rsrc(pooling ? p.pool + (size_t)b * p.Cout * PPL : nullptr, pooling ? p.Cout * PPL * 4 : 0);
    const __amdgpu_buffer_rsrc_t r_mask =
        mk_rsrc(p.mask_out ? p.mask_out + (size_t)b * p.Cout * PPL : nullptr, p.mask_out ? p.Cout * PPL : 0);
    const bool relu1 = p.relu && !p.add, relu2 = p.relu && p.add;   // (with a skip-add the ReLU comes
                                                                    //  after the sum)
    // store pass: thread -> (channel cl = 4 k + wave of the 32-channel block, row, 4-pixel piece)
    const int s_row = lane >> 3, s_x4 = (lane & 7) * 4;
    const int s_wy = wy0 + s_row, s_wx = wx0 + s_x4;
    const int s_nv = min(4, p.OW - s_wx);
    const bool s_ok = s_wy < p.OH && s_nv > 0;
    const unsigned s_out0 = 4u * (unsigned)((p.out_c0 + m0 + wave) * OPL + (p.out_y0 + s_wy) * p.out_W +
                                            p.out_x0 + s_wx);
    const unsigned s_add0 = 4u * (unsigned)((m0 + wave) * APL + (p.ay0 + s_wy) * p.AW + p.ax0 + s_wx);
    // pool pass: thread -> (channel cl = 4 k + wave, pooled row, pooled column) of the staged tile
    const int q_row = lane >> 4, q_col = lane & 15;
    const int q_wy = wy0 + 2 * q_row, q_wx = wx0 + 2 * q_col;
    const int q_py = (p.oy0 + q_wy) >> 1, q_px = (p.ox0 + q_wx) >> 1;
    const bool q_ok = pooling && q_wy + 1 < p.OH && q_wx + 1 < p.OW && q_py < p.pool_H && q_px < p.pool_W;
    const unsigned q_off0 = (unsigned)((m0 + wave) * PPL + q_py * p.pool_W + q_px);   // elements
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // bias of the 16 channels this lane holds in the MFMA C/D layout (0 beyond Cout / without bias)
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[r] = buf_ld(r_bias, 4u * (unsigned)(m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh), 0);
        __syncthreads();                       // previous users of these LDS bytes are done
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bv[r];
                if (relu1) v = fmaxf(v, 0.f);
                Cs[(cl * TH + lrow + j) * 32 + l31] = v;
            }
        __syncthreads();
        // 32 channels x TH rows x 8 pieces of 4 pixels: 8 passes, pass k handles channels 4k..4k+3
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cl = 4 * k + wave;
            const bool ok = s_ok && m0 + i * 32 + cl < p.Cout;
            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + (cl * TH + s_row) * 32 + s_x4);
            const unsigned oo = s_out0 + 4u * (unsigned)((i * 32 + 4 * k) * OPL);
            const unsigned ao = s_add0 + 4u * (unsigned)((i * 32 + 4 * k) * APL);
            if (p.add) {
                const f32x4 a4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_add, (int)((ok && s_nv == 4) ? ao : OOB), 0, 0));
                v += a4;
            }
            if (ok && s_nv < 4) {              // ragged right edge of the window: element by element
                for (int e = 0; e < s_nv; ++e) {
                    float t = v[e];
                    if (p.add) t += buf_ld(r_add, ao + 4u * e, 0);
                    if (relu2) t = fmaxf(t, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, t), r_out,
                                                          (int)(oo + 4u * e), 0, 0);
                }
            }
            if (relu2) {
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f);
                v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out,
                                                   (int)((ok && s_nv == 4) ? oo : OOB), 0, 0);
        }
        if (pooling) {
            // fused 2x2 max-pool of this 32-channel block from the staged tile: window origin and TH
            // are even, so the pairs (2m, 2m+1) of rows / columns are whole inside the tile; a
            // trailing unpaired row / column of the map has no pooling window (ignore_border)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int cl = 4 * k + wave;
                const bool ok = q_ok && m0 + i * 32 + cl < p.Cout;
                const float* c0 = Cs + (cl * TH + 2 * q_row) * 32 + 2 * q_col;
                const float m = fmaxf(fmaxf(c0[0], c0[1]), fmaxf(c0[32], c0[33]));
                const unsigned po = q_off0 + (unsigned)((i * 32 + 4 * k) * PPL);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, m), r_pool,
                                                      (int)(ok ? 4u * po : OOB), 0, 0);
                // bit (row & 1) * 2 + (col & 1): pre == pooled  (no descriptor records without mask_out)
                const unsigned bits = (c0[0] == m ? 1u : 0u) | (c0[1] == m ? 2u : 0u) |
                                      (c0[32] == m ? 4u : 0u) | (c0[33] == m ? 8u : 0u);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, r_mask, (int)(ok ? po : OOB), 0, 0);
            }
        }
    }
}

template <int BM, int TH, int WM, int WN>
int launch_halo_bf16(hipStream_t s, const ConvParams& cp, bool unpool) {
    ConvParams p = cp;
    const int tiles_y = (p.OH + TH - 1) / TH, tiles_x = (p.OW + 31) / 32;
    p.n_ptiles = p.B * tiles_y * tiles_x;
    p.n_mtiles = p.Mpad / BM;
    const int grid = p.n_ptiles * p.n_mtiles;
    if (unpool && p.mask_in)
        IISEG_LAUNCH((conv_halo_bf16_kernel<BM, TH, WM, WN, true, true>), dim3(grid), dim3(256),
                           0, s, p, tiles_y, tiles_x);
    else if (unpool)
        IISEG_LAUNCH((conv_halo_bf16_kernel<BM, TH, WM, WN, true>), dim3(grid), dim3(256), 0, s,
                           p, tiles_y, tiles_x);
    else
        IISEG_LAUNCH((conv_halo_bf16_kernel<BM, TH, WM, WN, false>), dim3(grid), dim3(256), 0, s,
                           p, tiles_y, tiles_x);
    return iiseg_check_launch();
}

int halo_bf16_bm(int Cout) { return Cout > 32 ? 64 : 32; }

int halo_bf16_check(const iiseg_conv_desc* d) {
    if (!d) return IISEG_ERR_NULL;
    if (d->KH != 3 || d->KW != 3 || d->dil != 1 || (d->flags & IISEG_CONV_TRANSPOSED2))
        return IISEG_ERR_UNSUPPORTED;
    if (d->B <= 0 || d->C1 <= 0 || d->C2 < 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 ||
        d->pad < 0 || d->OH <= 0 || d->OW <= 0 || d->oy0 < 0 || d->ox0 < 0)
        return IISEG_ERR_SHAPE;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (d->oy0 + d->OH > fullH || d->ox0 + d->OW > fullW) return IISEG_ERR_SHAPE;
    if ((d->flags & IISEG_CONV_UNPOOL) && d->C2 != 0) return IISEG_ERR_UNSUPPORTED;
    if (d->C2 > 0 && d->C1 % CPT) return IISEG_ERR_UNSUPPORTED;  // a k-tile must not straddle sources
    if (d->out_ctot != 0 && (d->out_c0 < 0 || d->out_c0 + d->Cout > d->out_ctot)) return IISEG_ERR_SHAPE;
    if (d->out_H != 0 && (d->out_y0 < 0 || d->out_x0 < 0 || d->out_y0 + d->OH > d->out_H ||
                          d->out_x0 + d->OW > d->out_W))
        return IISEG_ERR_SHAPE;
    const int64_t cmax = d->C1 > d->C2 ? d->C1 : d->C2;
    if (cmax * d->H * d->W * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;  // 32-bit byte offsets
    const int nkt = (d->C1 + d->C2 + CPT - 1) / CPT;
    const int bm = halo_bf16_bm(d->Cout), mpad = (d->Cout + bm - 1) / bm * bm;
    if ((int64_t)nkt * 18 * mpad * 16 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    // the epilogue addresses one image's output / skip-add / pooled planes with 32-bit byte offsets
    const int64_t octot = d->out_ctot ? d->out_ctot : d->Cout;
    const int64_t opl = d->out_H ? (int64_t)d->out_H * d->out_W : (int64_t)d->OH * d->OW;
    if ((octot + 64) * opl * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if (((int64_t)d->Cout + 64) * d->AH * d->AW * 4 >= (1ll << 31)) return IISEG_ERR_UNSUPPORTED;
    if ((int64_t)d->B * ((d->OH + 3) / 4) * ((d->OW + 31) / 32) * (mpad / 32) >= (1ll << 31))
        return IISEG_ERR_UNSUPPORTED;
    return IISEG_OK;
}

}  // namespace

extern "C" int iiseg_conv_halo_bf16_supported(const iiseg_conv_desc* d) {
    return halo_bf16_check(d) == IISEG_OK ? 1 : 0;
}

extern "C" int64_t iiseg_conv_halo_bf16_weight_bytes(const iiseg_conv_desc* d) {
    if (halo_bf16_check(d) != IISEG_OK) return 0;
    const int nkt = (d->C1 + d->C2 + CPT - 1) / CPT;
    const int bm = halo_bf16_bm(d->Cout), mpad = (d->Cout + bm - 1) / bm * bm;
    return (int64_t)nkt * 18 * mpad * 16;
}

extern "C" int iiseg_conv_halo_bf16_pack(void* stream, const iiseg_conv_desc* d, const float* w,
                                         int64_t stride_o, int64_t stride_c, void* wp16) {
    const int st = halo_bf16_check(d);
    if (st) return st;
    if (!w || !wp16) return IISEG_ERR_NULL;
    if ((uintptr_t)wp16 & 15) return IISEG_ERR_ALIGN;
    const int nkt = (d->C1 + d->C2 + CPT - 1) / CPT;
    const int bm = halo_bf16_bm(d->Cout), mpad = (d->Cout + bm - 1) / bm * bm;
    const int64_t n = (int64_t)nkt * 18 * mpad * 8;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    IISEG_LAUNCH(halo_pack_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_c, (__bf16*)wp16, d->C1 + d->C2, d->Cout, nkt, mpad);
    return iiseg_check_launch();
}

extern "C" int iiseg_conv_halo_bf16(void* stream, const iiseg_conv_desc* d, const float* x1,
                                    const float* x2, const float* pre, const float* pooled,
                                    const void* wp16, const float* bias, const float* add, float* out,
                                    float* pool_out, const unsigned char* mask_in,
                                    unsigned char* mask_out) {
    const int st = halo_bf16_check(d);
    if (st) return st;
    if (!x1 || !wp16) return IISEG_ERR_NULL;
    if (!out && !(pool_out && mask_out)) return IISEG_ERR_NULL;   // pre-pool map may be skipped
    if (mask_out && !pool_out) return IISEG_ERR_UNSUPPORTED;
    if (d->C2 > 0 && !x2) return IISEG_ERR_NULL;
    if ((uintptr_t)wp16 & 15) return IISEG_ERR_ALIGN;
    const bool unpool = (d->flags & IISEG_CONV_UNPOOL) != 0;
    if (unpool && !mask_in && (!pre || !pooled)) return IISEG_ERR_NULL;
    if (mask_in && !unpool) return IISEG_ERR_UNSUPPORTED;
    if (add && (d->AH < d->ay0 + d->OH || d->AW < d->ax0 + d->OW || d->ay0 < 0 || d->ax0 < 0))
        return IISEG_ERR_SHAPE;
    const int fullH = d->H + 2 * d->pad - 2, fullW = d->W + 2 * d->pad - 2;
    if (pool_out) {
        // whole pooling windows only: even origin, even extent unless the window ends at the map's
        // last (unpaired) row / column; no skip-add together with the pool
        if (add || ((d->oy0 | d->ox0) & 1) || ((d->OH & 1) && d->oy0 + d->OH != fullH) ||
            ((d->OW & 1) && d->ox0 + d->OW != fullW))
            return IISEG_ERR_UNSUPPORTED;
    }
    const int bm = halo_bf16_bm(d->Cout);
    ConvParams p = {};
    p.x1 = x1; p.x2 = x2; p.pre = pre; p.pooled = pooled;
    p.wp = (const float*)wp16;
    p.bias = bias; p.add = add; p.out = out;
    p.B = d->B; p.C1 = d->C1; p.C2 = d->C2; p.H = d->H; p.W = d->W;
    p.h2 = d->H / 2; p.w2 = d->W / 2;
    p.Cout = d->Cout; p.OH = d->OH; p.OW = d->OW; p.oy0 = d->oy0; p.ox0 = d->ox0;
    p.AH = d->AH; p.AW = d->AW; p.ay0 = d->ay0; p.ax0 = d->ax0;
    p.Kpad = (d->C1 + d->C2 + CPT - 1) / CPT;            // number of k-tiles
    p.Mpad = (d->Cout + bm - 1) / bm * bm;
    p.pad = d->pad; p.dil = 1;
    p.pool = pool_out;
    p.mask_in = mask_in; p.mask_out = mask_out;
    p.pool_H = fullH / 2; p.pool_W = fullW / 2;
    p.out_ctot = d->out_ctot ? d->out_ctot : d->Cout;
    p.out_c0 = d->out_ctot ? d->out_c0 : 0;
    p.out_H = d->out_H ? d->out_H : d->OH;
    p.out_W = d->out_H ? d->out_W : d->OW;
    p.out_y0 = d->out_H ? d->out_y0 : 0;
    p.out_x0 = d->out_H ? d->out_x0 : 0;
    p.P = d->B * d->OH * d->OW;
    p.relu = (d->flags & IISEG_CONV_RELU) ? 1 : 0;
    static const int dbg = getenv("IISEG_BF16_DEBUG") ? atoi(getenv("IISEG_BF16_DEBUG")) : 0;
    p.debug_nogather = dbg;
    hipStream_t s = (hipStream_t)stream;
    if (bm == 64) return launch_halo_bf16<64, 8, 1, 4>(s, p, unpool);
    return launch_halo_bf16<32, 8, 1, 4>(s, p, unpool);
}
