// Host-side pieces shared by the C8 convolution kernels (conv_c8_bf16.hip, conv_c8_m16.hip).
#pragma once
#include <stdint.h>

namespace iiseg {

// ceil(2^20 / d): n / d == (n * magic) >> 20 for n < 2^20 / d
inline unsigned magic20(int d) { return (unsigned)(((1u << 20) + (unsigned)d - 1) / (unsigned)d); }

// RECT tile shape for a window: th x tw <= cap pixels (256 or 512) whose (th + 2) x (tw + 2) patch fits
// `pcap` chunks, fewest tiles first, then the widest rows (longer store runs).  `even`: both even (quad order).
inline void rect_shape(int OH, int OW, int cap, int pcap, bool even, int* th_out, int* tw_out,
                       int64_t* tiles_out) {
    int64_t best = -1;
    int bth = 0, btw = 0;
    const int stepw = even ? 2 : 1;
    const int twmax = ((OW + stepw - 1) / stepw) * stepw;
    // (rows of at least 16 pixels where the window has them: 256-byte store runs)
    for (int tw = (twmax < 16 ? twmax : 16); tw <= cap && tw <= twmax; tw += stepw) {
        int th = cap / tw;
        if (even) th &= ~1;
        const int thmax = even ? ((OH + 1) & ~1) : OH;
        if (th > thmax) th = thmax;
        while (th > 0 && (th + 2) * (tw + 2) > pcap) th -= stepw;
        if (th <= 0) continue;
        const int rows = (OH + th - 1) / th, cols = (OW + tw - 1) / tw;
        const int64_t tiles = (int64_t)rows * cols;
        // the smallest tile that still makes rows x cols tiles (a smaller patch to stage)
        th = (OH + rows - 1) / rows;
        int tw_b = (OW + cols - 1) / cols;
        if (even) { th = (th + 1) & ~1; tw_b = (tw_b + 1) & ~1; }
        const int tw_k = tw;
        tw = tw_b;
        // (ties: the smallest patch -- measured: 16 x 32 runs 9 % faster than 8 x 62 at equal tile count --
        // then the wider rows: 6 x 40 beats 40 x 6 by 9 %, store runs)
        const int patch = (th + 2) * (tw + 2), bpatch = (bth + 2) * (btw + 2);
        if (best < 0 || tiles < best || (tiles == best && (patch < bpatch || (patch == bpatch && tw > btw)))) {
            best = tiles; bth = th; btw = tw;
        }
        tw = tw_k;
    }
    *th_out = bth; *tw_out = btw; *tiles_out = best;
}


}  // namespace iiseg
