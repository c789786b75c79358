// Declarations shared by the Winograd F(2x2,3x3) kernels of the 128 -> 128 channel 3x3 layer (conv3x3_wino.hip: 32 x 32 jobs,
// K-split, 16 x 16 jobs; conv3x3_wino_tn.hip: 16 channels x NB x 16 tiles per wave).  reference: code/autoencoder.py:274-287.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define WN_C 128
#define WN_FRAG_FLOATS (16 * WN_C * WN_C)          // one layout of the transformed filter
#define WN_PACKED_FLOATS (2 * WN_FRAG_FLOATS)      // [32-channel-tile fragments | 16-channel-tile fragments]
#define WN_OOB 0x80000000u        // byte offset beyond any image (128 H W 4 < 2^31): the buffer load returns 0, a store is dropped

struct WnArgs {
    const float* x; const float* wp; const float* scale; const float* shift;
    const float* res1; const float* res2; float* y;
    int N, H, W, grows, gcols, relu;
    int xcd_runs;               // 1 = contiguous runs of tile groups per XCD
    int g0;                     // first tile group of this launch (a shape may be split into launches of different forms)
    int ngroups;                // tile groups of this launch (the NB-segment kernels: 2 segments per group)
    int store_wt;               // NB-segment kernels: 1 = write-through output stores (single-round launches)
    unsigned mg_cols, mg_rows;  // NB-segment kernels: 2^32 / gcols + 1, 2^32 / grows + 1 (0 for a divisor of 1); set by the launcher
    unsigned long long* prof;   // profiling builds (WN_PROF) only
    const float* wp_next;       // F(4x4) kernel: the NEXT layer's fragments to pull into this XCD's L2 while this layer runs (nullptr: none)
    float* stats;               // F(4x4) kernel, STATS instantiations: per-segment (sum, sum of squares) of every output channel, [C][ngroups][2]
};

// conv3x3_wino_stack.hip: a whole residual stack (<= WN_STACK_MAX_LAYERS convs on one shape) as one persistent launch.
// The layer table travels in the kernel arguments (scalar loads by layer index, no table in device memory).
#define WN_STACK_MAX_LAYERS 40
#define WN_STACK_SYNC_BYTES 4096          // sync area in the caller's workspace: time-out word + <= 512 work-group flags
struct WnStackLayer {
    const float* x; const float* wp; const float* scale; const float* shift;
    const float* res1; const float* res2; float* y;
    int relu, pad_;
};
struct WnStackArgs {
    unsigned* flags;            // sync area: the time-out word, then layers completed per work-group
    int N, H, W, grows, gcols, nlayers, ngroups, xcd_runs;
    unsigned mg_cols, mg_rows, spin_limit, pad_;
    WnStackLayer layers[WN_STACK_MAX_LAYERS];
};

#ifdef __HIPCC__
// A tile row of 16 tiles is 16 lanes = one DPP row.
__device__ __forceinline__ float dpp_from_left(float edge, float v) {    // lane i <- v of lane i-1; row lane 0 keeps edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float edge, float v) {   // lane i <- v of lane i+1; row lane 15 keeps edge
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif

// conv3x3_wino4.hip: ic_wino4_3x3_c128_bn_act_f32 with the next layer's F(4x4) fragments to prefetch (nullptr: none)
int icx_wino4_3x3_c128_next(const float* x, const float* w_packed, const float* scale, const float* shift, const float* res1, const float* res2,
                            float* y, int N, int H, int W, int relu, int flags, const float* w_packed_next, hipStream_t st);
// conv3x3_wino.hip: ic_conv3x3_c128_auto_f32 with the next layer's `both` blob (nullptr: none)
int icx_conv3x3_c128_auto_next(const float* x, const float* w_both, const float* scale, const float* shift, const float* res1, const float* res2,
                               float* y, int N, int H, int W, int relu, int flags, const float* w_both_next, hipStream_t st);
// conv3x3_wino_tn.hip: launches the NB-segment kernel over tile groups [a.g0, a.g0 + a.ngroups); nb in {1, 2, 3}
int icx_wino_tn_launch(const WnArgs& a, int nb, int scalar_transform, hipStream_t st);
// conv3x3_wino_tp.hip: tile-pair / position-split jobs over tile groups [a.g0, a.g0 + a.ngroups), two work-groups per CU
int icx_wino_tp_launch(const WnArgs& a, hipStream_t st);
// conv3x3_wino_stack.hip: does the shape fit one resident round of nb-segment jobs / launch the persistent stack kernel
// (fills the geometry reciprocals, zeroes the flags on the stream first)
bool icx_wino_stack_fits(int N, int H, int W, int nb, int nlayers);
int icx_wino_stack_launch(WnStackArgs& a, int nb, hipStream_t st);
