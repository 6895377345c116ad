// Channel columns of NCHW tensors through buffer descriptors (tail.hip, metrics.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Channel columns through buffer descriptors.  One pixel per thread means C loads (stores) a plane apart per
// tensor; written as `if (c < C) v[c] = p[c * stride]` every load sat in a basic block of its own behind an
// s_waitcnt vmcnt(0) -- up to 33 dependent memory round trips per thread (refine_update: 0.125 ms for 526 MB).
// Here the offset of a channel past C, of a pixel past the map, or of a store to a stopped image is the
// out-of-range offset instead of a branch: all loads of a thread are issued back to back, the arithmetic
// follows, then the stores.  One descriptor per image and tensor (32-bit byte offsets: the host checks).
constexpr unsigned T_OOB = 0x80000000u;
typedef unsigned t_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t t_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00027000);
}
__device__ __forceinline__ void t_load(__amdgpu_buffer_rsrc_t r, unsigned off, float& v) {
    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void t_load(__amdgpu_buffer_rsrc_t r, unsigned off, double& v) {
    v = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ void t_store(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
}
__device__ __forceinline__ void t_store(__amdgpu_buffer_rsrc_t r, unsigned off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(t_u32x2, v), r, (int)off, 0, 0);
}
// v[c] = plane c of the tensor at byte offset off0 (channels >= C, or off0 out of range: 0, no memory access)
template <int CMAX, typename T>
__device__ __forceinline__ void load_column(__amdgpu_buffer_rsrc_t r, unsigned off0, unsigned plane_bytes, int C,
                                            T (&v)[CMAX]) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        t_load(r, (c < C && off0 != T_OOB) ? off0 + (unsigned)c * plane_bytes : T_OOB, v[c]);
}
template <int CMAX, typename T>
__device__ __forceinline__ void store_column(__amdgpu_buffer_rsrc_t r, unsigned off0, unsigned plane_bytes, int C,
                                             const T (&v)[CMAX]) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        t_store(r, (c < C && off0 != T_OOB) ? off0 + (unsigned)c * plane_bytes : T_OOB, v[c]);
}

}  // namespace
