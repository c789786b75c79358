// Training-mode BatchNorm around the autoencoder convs, and the backward of the importance map + quantiser.
//   reference: code/autoencoder.py:106-125 (slim.batch_norm, is_training=True: batch statistics over (N,H,W), biased
//   variance, eps 1e-5, decay 0.9), :127-134 (_quantize, qbar = qsoft + stop_gradient(qhard - qsoft)),
//   :171-200 (heatmap), code/quantizer.py:43-100.
// All of it is HBM streaming.  Per-channel sums are reduced in float64 in a FIXED order (deterministic, no atomics): BN_SPLIT
// work-groups per channel each sum a slice of the batch (per-thread double accumulators, LDS tree) into the caller's workspace,
// and whoever needs the totals adds the BN_SPLIT partials in index order -- the element-wise kernel that applies them does it
// itself in its prologue (a handful of loads), so a layer's forward is two launches and so is its backward:
//
// forward :  raw = conv(x)                      (conv kernels with scale = 1, shift = 0)
//            ic_bn_train_forward_f32(raw)       partial sums; then per plane: totals -> mean, var -> scale = gamma / sqrt(var + eps),
//                                               shift = beta - mean * scale -> y = act(raw * scale + shift) + res1 + res2
//                                               (one designated work-group per channel writes the statistics and the moving averages)
// backward:  g = dy * [raw * scale + shift > 0]  (ReLU mask recomputed, the residual adds pass dy through unchanged)
//            ic_bn_backward_f32                 partial sums of (g, g xhat); then per plane: totals (= dbeta, dgamma) ->
//                                               draw = gamma * invstd * (g - sum_g / M - xhat * sum_gxhat / M)
// Round 5: one 1024-thread work-group per channel (rounds 2-4) used 128 of the 256 CUs and was bound by what one CU can pull
// (7.3 / 10.4 us per layer for 16.8 / 33.5 MB); four work-groups per channel and 16-byte accesses in the element-wise kernels
// took the four BatchNorm launches of a 32 x 128 x 32 x 32 layer from 35.0 to the figures in DESIGN.md section 3 (training step).
#include "common.h"

#define BN_CHUNKS 64      // workspace layout: [C][BN_CHUNKS][2] partials (BN_SPLIT of the BN_CHUNKS slots are used), then [2][C] sums
#define BN_SPLIT 4        // work-groups per channel in the reductions

struct BnArgs {
    const float* x; const float* dy; const float* scale; const float* shift;
    const float* mean; const float* invstd; const float* gamma;
    const double* sums;       // [2][C] (sum_g, sum_gxhat); nullptr: take the totals from `partial`
    double* partial;          // workspace
    float* out0; float* out1; // stats: mean, var ; bwd_apply: dx
    int N, C, HW, relu;
    long long count;          // bwd_apply: elements per channel the sums run over (0 = N * HW; larger under sync BatchNorm)
    double* sums_out;         // bwd_apply with partials: where the designated work-group of a channel leaves the totals (or nullptr)
    float* dbeta; float* dgamma;
};

__device__ __forceinline__ void bn_block_reduce2_1024(double& a, double& b) {
    __shared__ double sa[1024], sb[1024];
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sa[threadIdx.x] += sa[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
        __syncthreads();
    }
    a = sa[0]; b = sb[0];
}

// Sums of channel c over the images [n0, n0 + nn) by one 1024-thread work-group.
// MODE 0: (sum x, sum x^2)   MODE 1: (sum g, sum g * xhat), g = dy masked by the ReLU of the forward pass
template <int MODE>
__device__ __forceinline__ void bn_channel_sums(const BnArgs& a, int c, int n0, int nn, double& s0, double& s1) {
    float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (MODE == 1) { sc = a.scale[c]; sh = a.shift[c]; mu = a.mean[c]; is = a.invstd[c]; }
    s0 = 0.0; s1 = 0.0;
    auto acc = [&](float xv, float g) __attribute__((always_inline)) {
        if (MODE == 0) { s0 += xv; s1 += (double)xv * xv; }
        else {
            if (a.relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
            s0 += g; s1 += (double)g * ((xv - mu) * is);
        }
    };
    if ((a.HW & 3) == 0) {
        // flattened float4 index i = n * HW4 + p4, stepped by 1024 with a carry instead of a division per load; four loads
        // in flight per thread
        const int HW4 = a.HW >> 2;
        const long long E4 = (long long)nn * HW4;
        const int dn = 1024 / HW4, dp = 1024 - dn * HW4;
        int n = (int)(threadIdx.x / HW4), p4 = (int)(threadIdx.x - n * HW4);
        long long i = threadIdx.x;
        auto step = [&]() __attribute__((always_inline)) { i += 1024; p4 += dp; n += dn; if (p4 >= HW4) { p4 -= HW4; ++n; } };
        auto offs = [&]() __attribute__((always_inline)) -> size_t { return ((size_t)(n0 + n) * a.C + c) * a.HW + 4 * (size_t)p4; };
        for (; i + 3 * 1024 < E4;) {
            float4 xv[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = offs();
                xv[u] = *reinterpret_cast<const float4*>(a.x + o);
                g[u] = MODE == 1 ? *reinterpret_cast<const float4*>(a.dy + o) : float4{0.f, 0.f, 0.f, 0.f};
                step();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc(xv[u].x, g[u].x); acc(xv[u].y, g[u].y); acc(xv[u].z, g[u].z); acc(xv[u].w, g[u].w); }
        }
        for (; i < E4; step()) {
            const size_t o = offs();
            const float4 xv = *reinterpret_cast<const float4*>(a.x + o);
            float4 g = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 1) g = *reinterpret_cast<const float4*>(a.dy + o);
            acc(xv.x, g.x); acc(xv.y, g.y); acc(xv.z, g.z); acc(xv.w, g.w);
        }
    } else {
        const long long E = (long long)nn * a.HW;
        for (long long i = threadIdx.x; i < E; i += 1024) {
            const int n = (int)(i / a.HW), p = (int)(i - (long long)n * a.HW);
            const size_t o = ((size_t)(n0 + n) * a.C + c) * a.HW + p;
            acc(a.x[o], MODE == 1 ? a.dy[o] : 0.f);
        }
    }
    bn_block_reduce2_1024(s0, s1);
}

// grid (BN_SPLIT, C): slice s of the batch of channel c -> partial[c][s]
template <int MODE>
__global__ __launch_bounds__(1024) void bn_partial_kernel(const BnArgs a) {
    const int s = blockIdx.x, c = blockIdx.y;
    const int per = (a.N + BN_SPLIT - 1) / BN_SPLIT;
    const int n0 = s * per < a.N ? s * per : a.N, n1 = n0 + per < a.N ? n0 + per : a.N;
    double s0, s1;
    bn_channel_sums<MODE>(a, c, n0, n1 - n0, s0, s1);
    if (threadIdx.x == 0) { a.partial[((size_t)c * BN_CHUNKS + s) * 2] = s0; a.partial[((size_t)c * BN_CHUNKS + s) * 2 + 1] = s1; }
}
// the totals of channel c: the partials in index order (the ONE place that defines the order -- every consumer calls this)
__device__ __forceinline__ void bn_totals(const double* __restrict__ partial, int c, double& s0, double& s1) {
    s0 = 0.0; s1 = 0.0;
#pragma unroll
    for (int s = 0; s < BN_SPLIT; ++s) { s0 += partial[((size_t)c * BN_CHUNKS + s) * 2]; s1 += partial[((size_t)c * BN_CHUNKS + s) * 2 + 1]; }
}

// what the forward pass derives from a channel's (sum x, sum x^2) over M elements: invstd and the folded scale / shift
__device__ __forceinline__ void bn_fold_values(double s0, double s1, long long M, float gamma, float beta, float eps,
                                               float& mf, float& vf, float& is, float& sc, float& sh) {
    const double m = s0 / (double)M;
    double v = s1 / (double)M - m * m;
    if (v < 0.0) v = 0.0;
    mf = (float)m; vf = (float)v;
    is = 1.0f / sqrtf(vf + eps);
    sc = gamma * is;
    sh = beta - mf * sc;
}
// ... and the moving-average update (decay 0.9; TF's fused kernel feeds the UNBIASED variance to the moving average while
// normalising with the biased one).  One function for the fused and the cross-replica path: the same bits from the same sums.
__device__ __forceinline__ void bn_fold_channel(int c, double s0, double s1, long long M, const float* gamma, const float* beta,
                                                float* moving_mean, float* moving_var, float decay, float eps, float* mean,
                                                float* invstd, float* scale, float* shift) {
    float mf, vf, is, sc, sh;
    bn_fold_values(s0, s1, M, gamma[c], beta[c], eps, mf, vf, is, sc, sh);
    mean[c] = mf; invstd[c] = is; scale[c] = sc; shift[c] = sh;
    if (moving_mean) moving_mean[c] = moving_mean[c] * decay + mf * (1.f - decay);
    if (moving_var) {
        const float unbiased = vf * (float)((double)M / (double)(M > 1 ? M - 1 : 1));
        moving_var[c] = moving_var[c] * decay + unbiased * (1.f - decay);
    }
}

struct BnFoldArgs {
    const float* gamma; const float* beta; float* moving_mean; float* moving_var; float decay, eps;
    float* mean; float* invstd; float* scale; float* shift;
};

// one thread per channel: the totals, and what the caller asked to be made of them
// OUT 0: mean / biased variance (ic_bn_stats_f32)   1: the sums as doubles (cross-replica paths; + dbeta / dgamma)   2: the whole fold
template <int OUT>
__global__ __launch_bounds__(64) void bn_finish_kernel(const double* __restrict__ partial, int C, long long M, double* __restrict__ sums,
                                                       float* __restrict__ out0, float* __restrict__ out1, const BnFoldArgs f) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s0, s1;
    bn_totals(partial, c, s0, s1);
    if (OUT == 0) {
        const double m = s0 / (double)M;
        double v = s1 / (double)M - m * m;
        if (v < 0.0) v = 0.0;
        out0[c] = (float)m; out1[c] = (float)v;
    } else if (OUT == 1) {
        sums[c] = s0; sums[C + c] = s1;
        if (out0) out0[c] = (float)s0;          // dbeta
        if (out1) out1[c] = (float)s1;          // dgamma
    } else {
        bn_fold_channel(c, s0, s1, M, f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay, f.eps, f.mean, f.invstd, f.scale, f.shift);
    }
}

// ---- element-wise halves: grid (plane chunks, C x image groups); a work-group serves BN_PL images of ONE channel (channel constants
// are block-uniform).  VEC (HW % 4 == 0, 16-byte aligned tensors): 16-byte accesses.  Measured on the cfg3 layer (32 x 128 x 32 x 32,
// 33.5 MB in + out, rocprofv3): scalar, one plane 8.5 us; 16-byte, one plane 8.9; 16-byte, 8 planes per work-group 10.3 -- all of them
// ~3.7 TB/s of read + write, which is what a device copy reaches on this chip: these passes are at the memory system's rate and only
// removing a pass (fusing the apply into the consumer's load) would make the layer's BatchNorm cheaper.  One plane per work-group. ----
#define BN_PL 1
struct BnApplyArgs {
    const float* x; const float* scale; const float* shift; const float* res1; const float* res2; float* y;
    int N, C, HW, relu;
    const double* partial;    // != nullptr: scale / shift are OUTPUTS -- folded here from the partial sums (BnFoldArgs f, count M)
    long long M;
    const float* cstats;      // != nullptr (instead of partial): the per-segment (sum, sum of squares) a convolution's epilogue left,
    int cparts;               //   [C][cparts][2] (ic_wino4_3x3_c128_raw_stats_f32) -- summed here in double, in index order
};
// the totals of channel c from a convolution's per-segment sums, by the work-group's first wave: lane l takes the segments l, l + 64, ...
// in double, a fixed xor tree over the 64 lanes, the result through LDS to the other waves -- the ONE place that defines this order
// (every work-group of the channel gets the same bits).  Called by ALL threads of the work-group (it holds a barrier).
// (Round 6, first version: every thread summed all segments itself -- 512 loads per thread in 4096 work-groups: 10 us per layer.)
__device__ __forceinline__ void bn_totals_cstats(const float* __restrict__ cs, int cparts, int c, double& s0, double& s1) {
    __shared__ double tot[2];
    if (threadIdx.x < 64) {
        const float* p = cs + (size_t)c * cparts * 2;
        double a0 = 0.0, a1 = 0.0;
        for (int i = threadIdx.x; i < cparts; i += 64) { a0 += (double)p[2 * i]; a1 += (double)p[2 * i + 1]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
        if (threadIdx.x == 0) { tot[0] = a0; tot[1] = a1; }
    }
    __syncthreads();
    s0 = tot[0]; s1 = tot[1];
}
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a, const BnFoldArgs f) {
    const int c = blockIdx.y % a.C;
    const int n_first = VEC ? (int)(blockIdx.y / a.C) * BN_PL : (int)(blockIdx.y / a.C);      // (grid.y = C x image groups)
    float sc, sh;
    if (a.partial || a.cstats) {
        double s0, s1;
        if (a.cstats) bn_totals_cstats(a.cstats, a.cparts, c, s0, s1);
        else bn_totals(a.partial, c, s0, s1);
        float mf, vf, is;
        bn_fold_values(s0, s1, a.M, f.gamma[c], f.beta[c], f.eps, mf, vf, is, sc, sh);
        // plane (n = 0, c), first chunk, first thread: the channel's statistics and moving averages, once
        if (n_first == 0 && blockIdx.x == 0 && threadIdx.x == 0)
            bn_fold_channel(c, s0, s1, a.M, f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay, f.eps, f.mean, f.invstd, f.scale, f.shift);
    } else { sc = a.scale[c]; sh = a.shift[c]; }
    const size_t base = ((size_t)n_first * a.C + c) * a.HW;
    if (VEC) {
        const int HW4 = a.HW >> 2;
        const size_t img4 = (size_t)a.C * HW4;                  // float4s from an image's plane to the next image's
        const int npl = a.N - n_first < BN_PL ? a.N - n_first : BN_PL;
        const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
        const float4* r1 = a.res1 ? reinterpret_cast<const float4*>(a.res1 + base) : nullptr;
        const float4* r2 = a.res2 ? reinterpret_cast<const float4*>(a.res2 + base) : nullptr;
        float4* y4 = reinterpret_cast<float4*>(a.y + base);
        for (int p = blockIdx.x * 256 + threadIdx.x; p < HW4; p += gridDim.x * 256) {
            float4 v[BN_PL];
#pragma unroll
            for (int j = 0; j < BN_PL; ++j) v[j] = j < npl ? x4[j * img4 + p] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < BN_PL; ++j) {
                if (j >= npl) break;
                float4 w = v[j];
                w.x = fmaf(w.x, sc, sh); w.y = fmaf(w.y, sc, sh); w.z = fmaf(w.z, sc, sh); w.w = fmaf(w.w, sc, sh);
                if (a.relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
                if (r1) { const float4 r = r1[j * img4 + p]; w.x += r.x; w.y += r.y; w.z += r.z; w.w += r.w; }
                if (r2) { const float4 r = r2[j * img4 + p]; w.x += r.x; w.y += r.y; w.z += r.z; w.w += r.w; }
                y4[j * img4 + p] = w;
            }
        }
    } else {
        for (int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) {
            float v = fmaf(a.x[base + p], sc, sh);
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.res1) v += a.res1[base + p];
            if (a.res2) v += a.res2[base + p];
            a.y[base + p] = v;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnArgs a) {
    const double M = a.count > 0 ? (double)a.count : (double)a.N * a.HW;
    const int c = blockIdx.y % a.C;
    const int n_first = VEC ? (int)(blockIdx.y / a.C) * BN_PL : (int)(blockIdx.y / a.C);      // (grid.y = C x image groups)
    const float sc = a.scale[c], sh = a.shift[c], mu = a.mean[c], is = a.invstd[c];
    double t0, t1;
    if (a.sums) { t0 = a.sums[c]; t1 = a.sums[a.C + c]; }
    else {
        bn_totals(a.partial, c, t0, t1);
        if (n_first == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
            if (a.sums_out) { a.sums_out[c] = t0; a.sums_out[a.C + c] = t1; }
            if (a.dbeta) a.dbeta[c] = (float)t0;
            if (a.dgamma) a.dgamma[c] = (float)t1;
        }
    }
    const float k = a.gamma[c] * is, mg = (float)(t0 / M), mgx = (float)(t1 / M);
    const size_t base = ((size_t)n_first * a.C + c) * a.HW;
    auto one = [&](float xv, float g) __attribute__((always_inline)) -> float {
        if (a.relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
        return k * (g - mg - (xv - mu) * is * mgx);
    };
    if (VEC) {
        const int HW4 = a.HW >> 2;
        const size_t img4 = (size_t)a.C * HW4;
        const int npl = a.N - n_first < BN_PL ? a.N - n_first : BN_PL;
        const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
        const float4* g4 = reinterpret_cast<const float4*>(a.dy + base);
        float4* o4 = reinterpret_cast<float4*>(a.out0 + base);
        for (int p = blockIdx.x * 256 + threadIdx.x; p < HW4; p += gridDim.x * 256) {
            float4 xv[BN_PL], g[BN_PL];
#pragma unroll
            for (int j = 0; j < BN_PL; ++j) { xv[j] = j < npl ? x4[j * img4 + p] : float4{0.f, 0.f, 0.f, 0.f}; g[j] = j < npl ? g4[j * img4 + p] : float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < BN_PL; ++j) {
                if (j >= npl) break;
                o4[j * img4 + p] = float4{one(xv[j].x, g[j].x), one(xv[j].y, g[j].y), one(xv[j].z, g[j].z), one(xv[j].w, g[j].w)};
            }
        }
    } else {
        for (int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) a.out0[base + p] = one(a.x[base + p], a.dy[base + p]);
    }
}

static inline bool bn_vec_ok(int HW, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4) {
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    return (HW & 3) == 0 && al(p0) && al(p1) && al(p2) && al(p3) && al(p4);
}
static inline dim3 bn_ew_grid(int HW, int N, int C, bool vec) {
    int gx = ic_cdiv(HW, 1024);                   // elements of a plane per work-group of 256 threads: one float4 or four floats per thread
    return dim3(gx < 1 ? 1 : gx, vec ? C * ic_cdiv(N, BN_PL) : N * C);
}
static void bn_launch_apply(const BnApplyArgs& a, const BnFoldArgs& f, int N, hipStream_t st) {
    const bool vec = bn_vec_ok(a.HW, a.x, a.res1, a.res2, a.y, nullptr);
    if (vec) hipLaunchKernelGGL(bn_apply_kernel<true>, bn_ew_grid(a.HW, N, a.C, true), dim3(256), 0, st, a, f);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, bn_ew_grid(a.HW, N, a.C, false), dim3(256), 0, st, a, f);
}
static void bn_launch_bwd_apply(const BnArgs& a, hipStream_t st) {
    const bool vec = bn_vec_ok(a.HW, a.x, a.dy, a.out0, nullptr, nullptr);
    if (vec) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, bn_ew_grid(a.HW, a.N, a.C, true), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, bn_ew_grid(a.HW, a.N, a.C, false), dim3(256), 0, st, a);
}

extern "C" size_t ic_bn_workspace_bytes(int C) { return C > 0 ? ((size_t)C * BN_CHUNKS * 2 + 2 * (size_t)C) * sizeof(double) : 0; }

extern "C" int ic_bn_stats_f32(const float* x, float* mean, float* var, int N, int C, int HW, void* workspace,
                               ic_stream_t stream) {
    IC_CHECK_ARG(x && mean && var && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.N = N; a.C = C; a.HW = HW; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(bn_finish_kernel<0>, dim3(ic_cdiv(C, 64)), dim3(64), 0, st, a.partial, C, (long long)N * HW, (double*)nullptr, mean, var, BnFoldArgs{});
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_train_stats_f32(const float* x, const float* gamma, const float* beta, float* moving_mean,
                                     float* moving_var, float decay, float eps, float* mean, float* invstd, float* scale,
                                     float* shift, int N, int C, int HW, void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(x && gamma && beta && mean && invstd && scale && shift && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.N = N; a.C = C; a.HW = HW; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const BnFoldArgs f{gamma, beta, moving_mean, moving_var, decay, eps, mean, invstd, scale, shift};
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(bn_finish_kernel<2>, dim3(ic_cdiv(C, 64)), dim3(64), 0, st, a.partial, C, (long long)N * HW, (double*)nullptr, (float*)nullptr, (float*)nullptr, f);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// statistics + fold + moving averages + normalise / activate / residual adds of one layer in two launches (the training loop's
// forward): same values, bit for bit, as ic_bn_train_stats_f32 followed by ic_bn_apply_f32
extern "C" int ic_bn_train_forward_f32(const float* x, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                                       float decay, float eps, float* mean, float* invstd, float* scale, float* shift,
                                       const float* res1, const float* res2, float* y, int N, int C, int HW, int relu,
                                       void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(x && gamma && beta && mean && invstd && scale && shift && y && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.N = N; a.C = C; a.HW = HW; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    const BnFoldArgs f{gamma, beta, moving_mean, moving_var, decay, eps, mean, invstd, scale, shift};
    BnApplyArgs ap{x, nullptr, nullptr, res1, res2, y, N, C, HW, relu, a.partial, (long long)N * HW, nullptr, 0};
    bn_launch_apply(ap, f, N, st);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// The same layer when the convolution that produced x left its channel sums behind (ic_wino4_3x3_c128_raw_stats_f32: per segment, fp32):
// ONE launch -- no pass over x for the statistics.  The sums are fp32 over <= 256 values per segment and double across segments:
// mean and variance agree with the two-pass form to ~1e-7 relative (tests/test_gpu_training.py compares them).
extern "C" int ic_bn_train_forward_cstats_f32(const float* x, const float* conv_stats, int parts, const float* gamma, const float* beta,
                                              float* moving_mean, float* moving_var, float decay, float eps, float* mean, float* invstd,
                                              float* scale, float* shift, const float* res1, const float* res2, float* y, int N, int C,
                                              int HW, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(x && conv_stats && parts > 0 && gamma && beta && mean && invstd && scale && shift && y && N > 0 && C > 0 && HW > 0);
    const BnFoldArgs f{gamma, beta, moving_mean, moving_var, decay, eps, mean, invstd, scale, shift};
    BnApplyArgs ap{x, nullptr, nullptr, res1, res2, y, N, C, HW, relu, nullptr, (long long)N * HW, conv_stats, parts};
    bn_launch_apply(ap, f, N, (hipStream_t)stream);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- cross-replica ("sync") BatchNorm: the same two passes with the per-channel sums handed to the caller in between ----
// Data-parallel training splits the reference's one batch over the ranks; its BatchNorm normalises over the WHOLE batch
// (autoencoder.py:115-125, batch_size 30 on one device).  The caller all-reduces the float64 sums (2 C doubles per layer)
// between the two halves: forward  ic_bn_moments_f32 -> sum over ranks -> ic_bn_train_fold_moments_f32,
//                         backward ic_bn_backward_reduce_f32 -> sum over ranks -> ic_bn_backward_apply_f32.
// With one rank and no all-reduce the results are bit-identical to ic_bn_train_stats_f32 / ic_bn_backward_f32.
extern "C" int ic_bn_moments_f32(const float* x, double* sums, int N, int C, int HW, void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(x && sums && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.N = N; a.C = C; a.HW = HW; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(bn_finish_kernel<1>, dim3(ic_cdiv(C, 64)), dim3(64), 0, st, a.partial, C, (long long)N * HW, sums, (float*)nullptr, (float*)nullptr, BnFoldArgs{});
    IC_LAUNCH_CHECK();
    return IC_OK;
}

__global__ void bn_fold_moments_kernel(const double* __restrict__ sums, int C, long long M, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ moving_mean,
                                       float* __restrict__ moving_var, float decay, float eps, float* __restrict__ mean,
                                       float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    bn_fold_channel(c, sums[c], sums[C + c], M, gamma, beta, moving_mean, moving_var, decay, eps, mean, invstd, scale, shift);
}

extern "C" int ic_bn_train_fold_moments_f32(const double* sums, long long count, const float* gamma, const float* beta,
                                            float* moving_mean, float* moving_var, float decay, float eps, float* mean,
                                            float* invstd, float* scale, float* shift, int C, ic_stream_t stream) {
    IC_CHECK_ARG(sums && gamma && beta && mean && invstd && scale && shift && C > 0 && count > 0);
    hipLaunchKernelGGL(bn_fold_moments_kernel, dim3(ic_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, C, count, gamma, beta,
                       moving_mean, moving_var, decay, eps, mean, invstd, scale, shift);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_backward_reduce_f32(const float* dy, const float* x, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, double* sums, float* dgamma, float* dbeta,
                                         int N, int C, int HW, int relu, void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(dy && x && scale && shift && mean && invstd && sums && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.dy = dy; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
    a.N = N; a.C = C; a.HW = HW; a.relu = relu; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(bn_finish_kernel<1>, dim3(ic_cdiv(C, 64)), dim3(64), 0, st, a.partial, C, (long long)N * HW, sums, dbeta, dgamma, BnFoldArgs{});
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_backward_apply_f32(const float* dy, const float* x, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, const float* gamma, const double* sums,
                                        long long count, float* dx, int N, int C, int HW, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(dy && x && scale && shift && mean && invstd && gamma && sums && dx && N > 0 && C > 0 && HW > 0 && count > 0);
    BnArgs a{};
    a.x = x; a.dy = dy; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
    a.N = N; a.C = C; a.HW = HW; a.relu = relu; a.sums = sums; a.out0 = dx; a.count = count;
    bn_launch_bwd_apply(a, (hipStream_t)stream);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_apply_f32(const float* x, const float* scale, const float* shift, const float* res1,
                               const float* res2, float* y, int N, int C, int HW, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(x && scale && shift && y && N > 0 && C > 0 && HW > 0);
    BnApplyArgs ap{x, scale, shift, res1, res2, y, N, C, HW, relu, nullptr, 0, nullptr, 0};
    bn_launch_apply(ap, BnFoldArgs{}, N, (hipStream_t)stream);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_backward_f32(const float* dy, const float* x, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, const float* gamma, float* dx,
                                  float* dgamma, float* dbeta, int N, int C, int HW, int relu, void* workspace,
                                  ic_stream_t stream) {
    IC_CHECK_ARG(dy && x && scale && shift && mean && invstd && gamma && dx && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.dy = dy; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
    a.N = N; a.C = C; a.HW = HW; a.relu = relu;
    a.partial = (double*)workspace;
    a.sums = nullptr; a.sums_out = nullptr; a.dbeta = dbeta; a.dgamma = dgamma; a.out0 = dx;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<1>, dim3(BN_SPLIT, C), dim3(1024), 0, st, a);
    bn_launch_bwd_apply(a, st);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ------------------------------------------------------------------------------------------------
// importance map + quantiser backward.  One lane = one pixel (n, p), looping over the C channels, so the
// gradient of the shared heatmap channel z0 is a plain in-lane sum (no atomics).
//   in : bottleneck (N,C+1,h,w), centers (L), d_qbar (N,C,h,w), d_heatmap (N,C,h,w) or null
//   out: d_bottleneck (N,C+1,h,w), d_centers partial sums [gridDim.x][L] -> ic_... stage 2 sums them in order
// ------------------------------------------------------------------------------------------------
#define Q_MAX_L 16
__global__ __launch_bounds__(256) void heatmap_quantize_bwd_kernel(
        const float* __restrict__ bn, const float* __restrict__ centers, int L, float sigma,
        const float* __restrict__ d_qbar, const float* __restrict__ d_heatmap, float* __restrict__ d_bn,
        double* __restrict__ dc_partial, int N, int C, int hw, int heatmap_on) {
    float c[Q_MAX_L];
#pragma unroll
    for (int j = 0; j < Q_MAX_L; ++j) c[j] = j < L ? centers[j] : 0.f;
    double dc[Q_MAX_L];
#pragma unroll
    for (int j = 0; j < Q_MAX_L; ++j) dc[j] = 0.0;
    const long long npix = (long long)N * hw;
    const int CB = C + (heatmap_on ? 1 : 0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long n = i / hw;
        const int p = (int)(i - n * hw);
        const float* b = bn + n * (long long)CB * hw;
        float* db = d_bn + n * (long long)CB * hw;
        float z0 = 0.f, sg = 1.f;
        if (heatmap_on) { z0 = b[p]; sg = 1.0f / (1.0f + expf(-z0)); }
        float dz0 = 0.f;
        for (int ch = 0; ch < C; ++ch) {
            const size_t e = ((size_t)n * C + ch) * hw + p;
            const float zc = b[(size_t)(ch + (heatmap_on ? 1 : 0)) * hw + p];
            float m = 1.f, u = 0.5f;
            if (heatmap_on) { u = sg * (float)C - (float)ch; m = fmaxf(fminf(u, 1.0f), 0.0f); }
            const float z = m * zc;
            // phi = softmax(-sigma d), qsoft = sum phi c
            float lmax = -INFINITY;
            float d[Q_MAX_L];
#pragma unroll
            for (int j = 0; j < Q_MAX_L; ++j) if (j < L) { const float t = z - c[j]; d[j] = t * t; lmax = fmaxf(lmax, -sigma * d[j]); }
            float den = 0.f, phi[Q_MAX_L];
#pragma unroll
            for (int j = 0; j < Q_MAX_L; ++j) if (j < L) { phi[j] = expf(-sigma * d[j] - lmax); den += phi[j]; }
            float qs = 0.f, mz = 0.f;                 // qsoft, sum_k phi_k (z - c_k)
#pragma unroll
            for (int j = 0; j < Q_MAX_L; ++j) if (j < L) { phi[j] /= den; qs += phi[j] * c[j]; mz += phi[j] * (z - c[j]); }
            const float gq = d_qbar[e];
            float dq_dz = 0.f;
#pragma unroll
            for (int j = 0; j < Q_MAX_L; ++j) if (j < L) {
                dq_dz += c[j] * phi[j] * (-2.f * sigma) * ((z - c[j]) - mz);
                dc[j] += (double)gq * (phi[j] + 2.f * sigma * (z - c[j]) * phi[j] * (c[j] - qs));
            }
            const float dz = gq * dq_dz;
            db[(size_t)(ch + (heatmap_on ? 1 : 0)) * hw + p] = dz * m;
            if (heatmap_on) {
                const float dm = dz * zc + (d_heatmap ? d_heatmap[e] : 0.f);
                if (u >= 0.f && u <= 1.f) dz0 += dm;          // clip passes the gradient inside [0, 1]
            }
        }
        if (heatmap_on) db[p] = dz0 * (float)C * sg * (1.f - sg);
    }
    // block reduction of the centre gradients
    __shared__ double sh[256];
    for (int j = 0; j < L; ++j) {
        sh[threadIdx.x] = dc[j];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) dc_partial[(size_t)blockIdx.x * L + j] = sh[0];
        __syncthreads();
    }
}

__global__ void dcenters_stage2(const double* __restrict__ partial, int nblocks, int L, float* __restrict__ out) {
    const int j = threadIdx.x;
    if (j >= L) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * L + j];
    out[j] = (float)s;
}

#define QB_BLOCKS 512
extern "C" size_t ic_heatmap_quantize_bwd_workspace_bytes(int L) { return (size_t)QB_BLOCKS * (L > 0 ? L : 0) * sizeof(double); }

extern "C" int ic_heatmap_quantize_bwd_f32(const float* bottleneck, const float* centers, int L, float sigma,
                                           const float* d_qbar, const float* d_heatmap, float* d_bottleneck,
                                           float* d_centers, int N, int C, int h, int w, int heatmap_on,
                                           void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(bottleneck && centers && d_qbar && d_bottleneck && d_centers && workspace && N > 0 && C > 0 && h > 0 && w > 0);
    if (L < 1 || L > Q_MAX_L) return IC_ERR_UNSUPPORTED;
    const long long npix = (long long)N * h * w;
    long long g = (npix + 255) / 256;
    const int blocks = (int)(g > QB_BLOCKS ? QB_BLOCKS : g);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(heatmap_quantize_bwd_kernel, dim3(blocks), dim3(256), 0, st, bottleneck, centers, L, sigma, d_qbar,
                       d_heatmap, d_bottleneck, (double*)workspace, N, C, h * w, heatmap_on);
    hipLaunchKernelGGL(dcenters_stage2, dim3(1), dim3(64), 0, st, (const double*)workspace, blocks, L, d_centers);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
