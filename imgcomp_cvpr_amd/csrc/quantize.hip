// Importance map + scalar quantiser (HBM-streaming, one lane per symbol).
//   reference: code/autoencoder.py:171-200 (_get_heatmap3D, _mask_with_heatmap), :127-134 (_quantize),
//              code/quantizer.py:43-100 (_quantize1d, phi_times_centers), :5 (_HARD_SIGMA = 1e7)
// The reference materialises (B,C,m,L) distance / softmax tensors; here the L <= 16 centres sit in
// registers and everything is one pass: 4(C+1)/C bytes in, up to 4*5+8 bytes out per symbol.
//
// Bit-exactness contract for `symbols`: the reference takes argmax_j softmax(-1e7 * d_j) with
// d_j = square(abs(z - c_j)).  softmax is strictly monotone in its argument at these magnitudes
// (two different fp32 logits differ by >= 1 ulp of ~1e5..1e7, i.e. exp(difference) < 1), so the
// result is the FIRST index attaining max_j fl(-1e7f * fl((z - c_j)^2)) -- which is what is computed
// here, with the products kept un-fused so that the fp32 roundings are the reference's.
#include "common.h"

#define IC_MAX_L 16
#define HARD_SIGMA 1e7f

__device__ __forceinline__ void quantize_one(float z, const float* c, int L, float sigma,
                                             float& qsoft, float& qhard, int& sym) {
    float d[IC_MAX_L];
    float lmax_soft = -INFINITY, lmax_hard = -INFINITY;
    int best = 0;
#pragma unroll
    for (int j = 0; j < IC_MAX_L; ++j) {
        if (j < L) {
            const float t = fabsf(z - c[j]);
            d[j] = __fmul_rn(t, t);
            const float lh = __fmul_rn(-HARD_SIGMA, d[j]);
            if (lh > lmax_hard) { lmax_hard = lh; best = j; }
            lmax_soft = fmaxf(lmax_soft, __fmul_rn(-sigma, d[j]));
        }
    }
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int j = 0; j < IC_MAX_L; ++j) {
        if (j < L) {
            const float e = expf(__fmul_rn(-sigma, d[j]) - lmax_soft);
            den += e;
        }
    }
#pragma unroll
    for (int j = 0; j < IC_MAX_L; ++j) {
        if (j < L) {
            const float e = expf(__fmul_rn(-sigma, d[j]) - lmax_soft);
            num += __fmul_rn(e / den, c[j]);      // phi_soft * centers, then reduce_sum (quantizer.py:98-100)
        }
    }
    qsoft = num;
    qhard = c[best];
    sym = best;
}

// centres are a device array (a trainable variable in the reference): read them in-kernel.
__global__ __launch_bounds__(256) void quantize_dev_centers_kernel(
        const float* __restrict__ z, const float* __restrict__ centers, int L, float sigma,
        float* __restrict__ qsoft, float* __restrict__ qhard, int64_t* __restrict__ symbols, long long count) {
    float c[IC_MAX_L];
#pragma unroll
    for (int j = 0; j < IC_MAX_L; ++j) c[j] = j < L ? centers[j] : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        float qs, qh; int s;
        quantize_one(z[i], c, L, sigma, qs, qh, s);
        if (qsoft) qsoft[i] = qs;
        if (qhard) qhard[i] = qh;
        if (symbols) symbols[i] = s;
    }
}

__global__ __launch_bounds__(256) void heatmap_quantize_kernel(
        const float* __restrict__ bn, const float* __restrict__ centers, int L, float sigma,
        float* __restrict__ heatmap, float* __restrict__ zout, float* __restrict__ qsoft,
        float* __restrict__ qhard, float* __restrict__ qbar, int64_t* __restrict__ symbols,
        int N, int C, int hw) {
    float c[IC_MAX_L];
#pragma unroll
    for (int j = 0; j < IC_MAX_L; ++j) c[j] = j < L ? centers[j] : 0.f;
    const long long count = (long long)N * C * hw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % hw);
        const long long t = i / hw;
        const int ch = (int)(t % C);
        const long long n = t / C;
        const float* b = bn + n * (long long)(C + 1) * hw;
        const float z0 = b[p];
        const float zc = b[(long long)(ch + 1) * hw + p];
        // heatmap2D = sigmoid(z0) * C ; heatmap3D = max(min(h - c, 1), 0)   (autoencoder.py:182-194)
        const float h2 = __fmul_rn(1.0f / (1.0f + expf(-z0)), (float)C);
        const float m = fmaxf(fminf(h2 - (float)ch, 1.0f), 0.0f);
        const float z = __fmul_rn(m, zc);
        float qs, qh; int s;
        quantize_one(z, c, L, sigma, qs, qh, s);
        if (heatmap) heatmap[i] = m;
        if (zout) zout[i] = z;
        if (qsoft) qsoft[i] = qs;
        if (qhard) qhard[i] = qh;
        if (qbar) qbar[i] = qs + (qh - qs);      // forward value of qsoft + stop_gradient(qhard - qsoft)
        if (symbols) symbols[i] = s;
    }
}

static int grid_for(long long count) {
    long long g = (count + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" int ic_quantize_f32(const float* z, const float* centers, int L, float sigma,
                               float* qsoft, float* qhard, int64_t* symbols, long long count, ic_stream_t stream) {
    IC_CHECK_ARG(z && centers && count > 0);
    if (L < 1 || L > IC_MAX_L) return IC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(quantize_dev_centers_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream,
                       z, centers, L, sigma, qsoft, qhard, symbols, count);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_heatmap_quantize_f32(const float* bottleneck, const float* centers, int L, float sigma,
                                       float* heatmap, float* z, float* qsoft, float* qhard, float* qbar,
                                       int64_t* symbols, int N, int C, int h, int w, ic_stream_t stream) {
    IC_CHECK_ARG(bottleneck && centers && N > 0 && C > 0 && h > 0 && w > 0);
    if (L < 1 || L > IC_MAX_L) return IC_ERR_UNSUPPORTED;
    const long long count = (long long)N * C * h * w;
    hipLaunchKernelGGL(heatmap_quantize_kernel, dim3(grid_for(count)), dim3(256), 0, (hipStream_t)stream,
                       bottleneck, centers, L, sigma, heatmap, z, qsoft, qhard, qbar, symbols, N, C, h * w);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
