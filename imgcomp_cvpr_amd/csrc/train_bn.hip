// Training-mode BatchNorm around the autoencoder convs, and the backward of the importance map + quantiser.
//   reference: code/autoencoder.py:106-125 (slim.batch_norm, is_training=True: batch statistics over (N,H,W), biased
//   variance, eps 1e-5, decay 0.9), :127-134 (_quantize, qbar = qsoft + stop_gradient(qhard - qsoft)),
//   :171-200 (heatmap), code/quantizer.py:43-100.
// All of it is HBM streaming: every kernel reads/writes each activation once, per-channel sums are reduced in
// float64 in a fixed order by ONE work-group per channel (deterministic, no atomics).
//
// forward :  raw = conv(x)                      (conv kernels with scale = 1, shift = 0)
//            mean, var = ic_bn_stats_f32(raw)   -> host folds scale = gamma / sqrt(var + eps), shift = beta - mean * scale
//            y = ic_bn_apply_f32(raw, scale, shift, relu, res1, res2)
// backward:  g = dy * [raw * scale + shift > 0]  (ReLU mask recomputed, the residual adds pass dy through unchanged)
//            ic_bn_bwd_reduce_f32 -> sum_g, sum_gxhat per channel  (= dbeta, dgamma)
//            ic_bn_bwd_apply_f32  -> draw = gamma * invstd * (g - sum_g / M - xhat * sum_gxhat / M)
#include "common.h"

#define BN_CHUNKS 64      // (workspace layout of ABI version 1: [C][BN_CHUNKS][2] partials, then the [2][C] sums still used)

struct BnArgs {
    const float* x; const float* dy; const float* scale; const float* shift;
    const float* mean; const float* invstd; const float* gamma;
    const double* sums;       // [2][C] (sum_g, sum_gxhat)
    double* partial;          // workspace
    float* out0; float* out1; // stats: mean, var ; bwd_apply: dx
    int N, C, HW, relu;
    long long count;          // bwd_apply: elements per channel the sums run over (0 = N * HW; larger under sync BatchNorm)
};

// ---- one work-group of 1024 threads per channel: sums, and what follows from them, in ONE launch ----------------------------
// The two-stage form of round 1 cost two launches per reduction (stage 1 over [C][chunks] blocks, stage 2 over the partials) and
// the forward pass a third for the fold -- 5-6 us each of mostly launch latency, 140 + 70 times per training step.  A channel
// of the training shapes is 128 K elements: one 1024-thread block streams it in a few microseconds (16-byte loads, 128
// blocks on 128 CUs), reduces in a fixed order (per-thread double accumulators, LDS tree) and finishes the job itself.
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

// MODE 0: (sum x, sum x^2)   MODE 1: (sum g, sum g * xhat), g = dy masked by the ReLU of the forward pass
template <int MODE>
__device__ __forceinline__ void bn_channel_sums(const BnArgs& a, int c, double& s0, double& s1) {
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
        // in flight per thread (the block is alone on its CU: memory-level parallelism has to come from within the thread)
        const int HW4 = a.HW >> 2;
        const long long E4 = (long long)a.N * HW4;
        const int dn = 1024 / HW4, dp = 1024 - dn * HW4;
        int n = (int)(threadIdx.x / HW4), p4 = (int)(threadIdx.x - n * HW4);
        long long i = threadIdx.x;
        auto step = [&]() __attribute__((always_inline)) { i += 1024; p4 += dp; n += dn; if (p4 >= HW4) { p4 -= HW4; ++n; } };
        auto offs = [&]() __attribute__((always_inline)) -> size_t { return ((size_t)n * a.C + c) * a.HW + 4 * (size_t)p4; };
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
        const long long E = (long long)a.N * a.HW;
        for (long long i = threadIdx.x; i < E; i += 1024) {
            const int n = (int)(i / a.HW), p = (int)(i - (long long)n * a.HW);
            const size_t o = ((size_t)n * a.C + c) * a.HW + p;
            acc(a.x[o], MODE == 1 ? a.dy[o] : 0.f);
        }
    }
    bn_block_reduce2_1024(s0, s1);
}

// what the forward pass derives from a channel's (sum x, sum x^2) over M elements: invstd, the folded scale/shift and the
// moving-average update (decay 0.9; TF's fused kernel feeds the UNBIASED variance to the moving average while normalising
// with the biased one).  One function for the fused and the cross-replica path: the same bits from the same sums.
__device__ __forceinline__ void bn_fold_channel(int c, double s0, double s1, long long M, const float* gamma, const float* beta,
                                                float* moving_mean, float* moving_var, float decay, float eps, float* mean,
                                                float* invstd, float* scale, float* shift) {
    const double m = s0 / (double)M;
    double v = s1 / (double)M - m * m;
    if (v < 0.0) v = 0.0;
    const float mf = (float)m, vf = (float)v;
    const float is = 1.0f / sqrtf(vf + eps);
    const float sc = gamma[c] * is;
    mean[c] = mf; invstd[c] = is; scale[c] = sc; shift[c] = beta[c] - mf * sc;
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

// OUT 0: mean / biased variance (ic_bn_stats_f32)   1: the sums as doubles (cross-replica path)   2: the whole fold
template <int OUT>
__global__ __launch_bounds__(1024) void bn_channel_stats_kernel(const BnArgs a, double* __restrict__ sums, const BnFoldArgs f) {
    const int c = blockIdx.x;
    double s0, s1;
    bn_channel_sums<0>(a, c, s0, s1);
    if (threadIdx.x != 0) return;
    const long long M = (long long)a.N * a.HW;
    if (OUT == 0) {
        const double m = s0 / (double)M;
        double v = s1 / (double)M - m * m;
        if (v < 0.0) v = 0.0;
        a.out0[c] = (float)m; a.out1[c] = (float)v;
    } else if (OUT == 1) {
        sums[c] = s0; sums[a.C + c] = s1;
    } else {
        bn_fold_channel(c, s0, s1, M, f.gamma, f.beta, f.moving_mean, f.moving_var, f.decay, f.eps, f.mean, f.invstd, f.scale, f.shift);
    }
}

// backward: (sum g, sum g xhat) -> sums[2][C] (doubles, for the data gradient) and dbeta / dgamma
__global__ __launch_bounds__(1024) void bn_channel_bwd_sums_kernel(const BnArgs a, double* __restrict__ sums, float* __restrict__ dbeta,
                                                                   float* __restrict__ dgamma) {
    const int c = blockIdx.x;
    double s0, s1;
    bn_channel_sums<1>(a, c, s0, s1);
    if (threadIdx.x != 0) return;
    sums[c] = s0; sums[a.C + c] = s1;
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
}

// grid (plane chunks, N*C planes): one (n, c) plane per blockIdx.y -> channel constants are block-uniform
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res1,
                                                       const float* __restrict__ res2, float* __restrict__ y,
                                                       int C, int HW, long long total, int relu) {
    const int c = blockIdx.y % C;
    const float sc = scale[c], sh = shift[c];
    const size_t base = (size_t)blockIdx.y * HW;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float v = fmaf(x[base + p], sc, sh);
        if (relu) v = fmaxf(v, 0.f);
        if (res1) v += res1[base + p];
        if (res2) v += res2[base + p];
        y[base + p] = v;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnArgs a, long long total) {
    const double M = a.count > 0 ? (double)a.count : (double)a.N * a.HW;
    const int c = blockIdx.y % a.C;
    const float sc = a.scale[c], sh = a.shift[c], mu = a.mean[c], is = a.invstd[c];
    const float k = a.gamma[c] * is, mg = (float)(a.sums[c] / M), mgx = (float)(a.sums[a.C + c] / M);
    const size_t base = (size_t)blockIdx.y * a.HW;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) {
        const float xv = a.x[base + p];
        float g = a.dy[base + p];
        if (a.relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
        a.out0[base + p] = k * (g - mg - (xv - mu) * is * mgx);
    }
}

static int ew_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" size_t ic_bn_workspace_bytes(int C) { return C > 0 ? ((size_t)C * BN_CHUNKS * 2 + 2 * (size_t)C) * sizeof(double) : 0; }

extern "C" int ic_bn_stats_f32(const float* x, float* mean, float* var, int N, int C, int HW, void* workspace,
                               ic_stream_t stream) {
    IC_CHECK_ARG(x && mean && var && workspace && N > 0 && C > 0 && HW > 0);
    BnArgs a{};
    a.x = x; a.N = N; a.C = C; a.HW = HW; a.partial = (double*)workspace;
    hipStream_t st = (hipStream_t)stream;
    a.out0 = mean; a.out1 = var;
    hipLaunchKernelGGL(bn_channel_stats_kernel<0>, dim3(C), dim3(1024), 0, st, a, (double*)nullptr, BnFoldArgs{});
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
    hipLaunchKernelGGL(bn_channel_stats_kernel<2>, dim3(C), dim3(1024), 0, st, a, (double*)nullptr, f);
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
    hipLaunchKernelGGL(bn_channel_stats_kernel<1>, dim3(C), dim3(1024), 0, st, a, sums, BnFoldArgs{});
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
    a.N = N; a.C = C; a.HW = HW; a.relu = relu;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_channel_bwd_sums_kernel, dim3(C), dim3(1024), 0, st, a, sums, dbeta, dgamma);
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
    const long long total = (long long)N * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ic_cdiv(HW, 1024) < 1 ? 1 : ic_cdiv(HW, 1024), N * C), dim3(256), 0,
                       (hipStream_t)stream, a, total);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_bn_apply_f32(const float* x, const float* scale, const float* shift, const float* res1,
                               const float* res2, float* y, int N, int C, int HW, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(x && scale && shift && y && N > 0 && C > 0 && HW > 0);
    const long long total = (long long)N * C * HW;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ic_cdiv(HW, 1024) < 1 ? 1 : ic_cdiv(HW, 1024), N * C), dim3(256), 0,
                       (hipStream_t)stream, x, scale, shift, res1, res2, y, C, HW, total, relu);
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
    double* sums = a.partial + (size_t)C * BN_CHUNKS * 2;
    a.sums = sums; a.out0 = dx;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_channel_bwd_sums_kernel, dim3(C), dim3(1024), 0, st, a, sums, dbeta, dgamma);
    const long long total = (long long)N * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ic_cdiv(HW, 1024) < 1 ? 1 : ic_cdiv(HW, 1024), N * C), dim3(256), 0, st, a, total);
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
