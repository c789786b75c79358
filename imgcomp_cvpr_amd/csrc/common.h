// Shared helpers for the gfx950 kernels of libimgcomp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/imgcomp_hip.h"

#define IC_CHECK_ARG(cond) do { if (!(cond)) return IC_ERR_ARG; } while (0)
#define IC_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

static inline int ic_cdiv(int a, int b) { return (a + b - 1) / b; }

// TensorFlow 'SAME' padding: out = ceil(in/stride); total = max((out-1)*stride + k - in, 0);
// pad_before = total / 2 (the odd pixel goes to the bottom/right).
static inline int ic_same_pad_before(int in, int k, int stride) {
    int out = (in + stride - 1) / stride;
    int total = (out - 1) * stride + k - in;
    if (total < 0) total = 0;
    return total / 2;
}

// reference: code/autoencoder.py:162-163 (mean, var) and :143 (sqrt(var + 1e-10), float32)
static __device__ __constant__ const float IC_IMG_MEAN[3] = {121.85369873f, 113.58860779f, 100.63715363f};
static __device__ __constant__ const float IC_IMG_STD[3] = {68.8939514f, 66.7393417f, 69.3702698f};
// Work-group i of a launch runs on XCD i % 8 (round-robin dispatch), each XCD with its own L2.  Kernels whose neighbouring
// tiles share input (halo rows, the channel tiles of one tile group) decode their tile from ic_xcd_run(blockIdx.x, gridDim.x)
// instead of blockIdx.x: every XCD then walks ONE contiguous run of tiles and finds its neighbours' data in its own L2.
// A bijection on [0, n): the first 8 * (n / 8) indices are permuted, the tail keeps its place.
#ifdef __HIPCC__
__device__ __forceinline__ int ic_xcd_run(int b, int n) {
    const int per = n >> 3;
    return b < 8 * per ? (b & 7) * per + (b >> 3) : b;
}
#endif

