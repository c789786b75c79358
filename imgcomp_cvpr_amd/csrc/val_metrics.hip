// val.py's image metrics on the device: MS-SSIM as the reference's numpy code computes it (code/ms_ssim_np.py:49-110 -- 5 scales, an
// 11-tap Gaussian blur in 'valid' mode, float64 throughout, a 2x2 box with the far edge repeated between scales) and the mean squared
// error behind PSNR (val.py:92-96), of two uint8 NCHW batches.  The torch float64 twin (imgcomp_cvpr_amd/metrics.py, still the CPU
// path) dispatches ~1000 element-wise kernels per Kodak image: 8.3 ms of host time against 2.4 ms for the whole codec path (round 4,
// tools/val_host_profile.py).  Here: per scale one vertical pass (5 blurred maps: x, y, x^2, y^2, xy), one horizontal pass that ends
// in the per-pixel SSIM / contrast-structure terms and per-block sums, one down-sampling launch; one launch for the squared error,
// one that adds the block sums in a fixed order.  Same operations in the same order per value as the numpy code (products and sums
// rounded separately: no fused multiply-add), so the per-scale means agree to ~1e-15.
#include "common.h"
#include "internal.h"

#define VM_SCALES 5
#define VM_MAXTAP 11

struct VmTaps { int size; double g[VM_MAXTAP]; };

template <typename T>
__global__ __launch_bounds__(256) void vm_vertical_kernel(const T* __restrict__ x, const T* __restrict__ y, double* __restrict__ tmp,
                                                         long long planes, int H, int W, int Ho, VmTaps tp) {
    const long long total = planes * Ho * W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % W);
    const long long t2 = idx / W;
    const int i = (int)(t2 % Ho);
    const long long p = t2 / Ho;
    const T* xp = x + (p * H + i) * W + j;
    const T* yp = y + (p * H + i) * W + j;
    double s[5];
#pragma unroll
    for (int t = 0; t < VM_MAXTAP; ++t) {
        if (t < tp.size) {
            const double a = (double)xp[(long long)t * W], b = (double)yp[(long long)t * W], g = tp.g[t];
            const double m[5] = {a, b, __dmul_rn(a, a), __dmul_rn(b, b), __dmul_rn(a, b)};
#pragma unroll
            for (int k = 0; k < 5; ++k) s[k] = t == 0 ? __dmul_rn(g, m[k]) : __dadd_rn(s[k], __dmul_rn(g, m[k]));
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) tmp[(k * planes + p) * (long long)Ho * W + (long long)i * W + j] = s[k];
}

// block sums of the per-pixel terms: partial[2 b] = sum of SSIM, partial[2 b + 1] = sum of contrast-structure
__global__ __launch_bounds__(256) void vm_horizontal_kernel(const double* __restrict__ tmp, double* __restrict__ partial,
                                                            long long planes, int Ho, int W, int Wo, VmTaps tp, double c1, double c2) {
    __shared__ double red[2][256];
    const long long total = planes * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    double ssim = 0.0, cs = 0.0;
    if (idx < total) {
        const int j = (int)(idx % Wo);
        const long long t2 = idx / Wo;                            // plane * Ho + row
        double o[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double* r = tmp + (k * planes * Ho + t2) * (long long)W + j;
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < VM_MAXTAP; ++t)
                if (t < tp.size) acc = t == 0 ? __dmul_rn(tp.g[0], r[0]) : __dadd_rn(acc, __dmul_rn(tp.g[t], r[t]));
            o[k] = acc;
        }
        const double mu1 = o[0], mu2 = o[1];
        const double s11 = __dsub_rn(o[2], __dmul_rn(mu1, mu1)), s22 = __dsub_rn(o[3], __dmul_rn(mu2, mu2)), s12 = __dsub_rn(o[4], __dmul_rn(mu1, mu2));
        const double v1 = __dadd_rn(__dmul_rn(2.0, s12), c2), v2 = __dadd_rn(__dadd_rn(s11, s22), c2);
        const double num = __dmul_rn(__dadd_rn(__dmul_rn(__dmul_rn(2.0, mu1), mu2), c1), v1);
        const double den = __dmul_rn(__dadd_rn(__dadd_rn(__dmul_rn(mu1, mu1), __dmul_rn(mu2, mu2)), c1), v2);
        ssim = num / den;
        cs = v1 / v2;
    }
    red[0][threadIdx.x] = ssim; red[1][threadIdx.x] = cs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}

// 0.25 * (p[i][j] + p[i + 1][j] + p[i][j + 1] + p[i + 1][j + 1]) at even i, j, the far edge repeated (np.pad 'symmetric' by one)
template <typename T>
__global__ __launch_bounds__(256) void vm_down_kernel(const T* __restrict__ x, const T* __restrict__ y, double* __restrict__ xo, double* __restrict__ yo,
                                                     long long planes, int H, int W, int H2, int W2) {
    const long long total = planes * H2 * W2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % W2);
    const long long t2 = idx / W2;
    const int i = (int)(t2 % H2);
    const long long p = t2 / H2;
    const int i0 = 2 * i, j0 = 2 * j, i1 = min(i0 + 1, H - 1), j1 = min(j0 + 1, W - 1);
    const T* xp = x + p * H * W;
    const T* yp = y + p * H * W;
    xo[idx] = __dmul_rn(0.25, __dadd_rn(__dadd_rn(__dadd_rn((double)xp[(long long)i0 * W + j0], (double)xp[(long long)i1 * W + j0]), (double)xp[(long long)i0 * W + j1]),
                                        (double)xp[(long long)i1 * W + j1]));
    yo[idx] = __dmul_rn(0.25, __dadd_rn(__dadd_rn(__dadd_rn((double)yp[(long long)i0 * W + j0], (double)yp[(long long)i1 * W + j0]), (double)yp[(long long)i0 * W + j1]),
                                        (double)yp[(long long)i1 * W + j1]));
}

__global__ __launch_bounds__(256) void vm_sqerr_kernel(const unsigned char* __restrict__ x, const unsigned char* __restrict__ y, double* __restrict__ partial,
                                                      long long total) {
    __shared__ double red[256];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const double d = (double)x[i] - (double)y[i];
        s += d * d;                                               // integers below 2^53: exact in any order
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct VmFinish { const double* partial[VM_SCALES + 1]; int blocks[VM_SCALES + 1]; int stride[VM_SCALES + 1]; int pick[VM_SCALES + 1]; double count[VM_SCALES + 1]; };
// out[s] = mean contrast-structure of scale s (s < 4), mean SSIM of scale 4, out[5] = mean squared error: block sums added in block order
__global__ __launch_bounds__(256) void vm_finish_kernel(VmFinish f, double* __restrict__ out) {
    __shared__ double red[256];
    for (int s = 0; s <= VM_SCALES; ++s) {
        double acc = 0.0;
        for (int b = threadIdx.x; b < f.blocks[s]; b += 256) acc += f.partial[s][(long long)b * f.stride[s] + f.pick[s]];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[s] = red[0] / f.count[s];
        __syncthreads();
    }
}

static VmTaps vm_taps(int h, int w) {                             // ms_ssim_np.py:35-46, :75-77: size = min(11, h, w), sigma = size * 1.5 / 11
    VmTaps t{};
    t.size = h < w ? (h < VM_MAXTAP ? h : VM_MAXTAP) : (w < VM_MAXTAP ? w : VM_MAXTAP);
    const double sigma = t.size * 1.5 / 11.0;
    const int r = t.size / 2;
    double sum = 0.0;
    for (int i = 0; i < t.size; ++i) {
        const double xx = (t.size % 2 == 0) ? (double)(i - r) + 0.5 : (double)(i - r);
        t.g[i] = exp(-(xx * xx) / (2.0 * sigma * sigma));
        sum += t.g[i];
    }
    for (int i = 0; i < t.size; ++i) t.g[i] /= sum;
    return t;
}

struct VmPlan { int H[VM_SCALES], W[VM_SCALES]; size_t plane_off[VM_SCALES]; size_t tmp_off, part_off[VM_SCALES + 1], doubles; int blocks[VM_SCALES + 1]; };
static VmPlan vm_plan(long long planes, int H, int W) {
    VmPlan p{};
    size_t off = 0;
    size_t tmp_max = 0;
    for (int s = 0; s < VM_SCALES; ++s) {
        p.H[s] = s ? (p.H[s - 1] + 1) / 2 : H;
        p.W[s] = s ? (p.W[s - 1] + 1) / 2 : W;
        if (s) { p.plane_off[s] = off; off += 2 * (size_t)planes * p.H[s] * p.W[s]; }
        const VmTaps t = vm_taps(p.H[s], p.W[s]);
        const size_t tm = 5 * (size_t)planes * (p.H[s] - t.size + 1) * p.W[s];
        if (tm > tmp_max) tmp_max = tm;
        p.blocks[s] = (int)(((size_t)planes * (p.H[s] - t.size + 1) * (p.W[s] - t.size + 1) + 255) / 256);
    }
    p.tmp_off = off; off += tmp_max;
    for (int s = 0; s < VM_SCALES; ++s) { p.part_off[s] = off; off += 2 * (size_t)p.blocks[s]; }
    p.blocks[VM_SCALES] = 1024;
    p.part_off[VM_SCALES] = off; off += 1024;
    p.doubles = off;
    return p;
}

extern "C" size_t ic_val_metrics_workspace_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return vm_plan((long long)N * C, H, W).doubles * sizeof(double);
}

extern "C" int ic_val_metrics_u8_f64(const unsigned char* x, const unsigned char* y, int N, int C, int H, int W, double* out6,
                                     void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(x && y && out6 && workspace && N > 0 && C > 0 && H > 0 && W > 0);
    const long long planes = (long long)N * C;
    const VmPlan p = vm_plan(planes, H, W);
    if (workspace_bytes < p.doubles * sizeof(double)) return IC_ERR_WORKSPACE;
    if (planes * H * W >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    double* ws = (double*)workspace;
    const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
    VmFinish f{};
    for (int s = 0; s < VM_SCALES; ++s) {
        const int h = p.H[s], w = p.W[s];
        const VmTaps t = vm_taps(h, w);
        const int ho = h - t.size + 1, wo = w - t.size + 1;
        double* tmp = ws + p.tmp_off;
        const unsigned gv = (unsigned)((planes * ho * w + 255) / 256);
        if (s == 0) hipLaunchKernelGGL(vm_vertical_kernel<unsigned char>, dim3(gv), dim3(256), 0, st, x, y, tmp, planes, h, w, ho, t);
        else hipLaunchKernelGGL(vm_vertical_kernel<double>, dim3(gv), dim3(256), 0, st, (const double*)(ws + p.plane_off[s]),
                                (const double*)(ws + p.plane_off[s] + (size_t)planes * h * w), tmp, planes, h, w, ho, t);
        hipLaunchKernelGGL(vm_horizontal_kernel, dim3((unsigned)p.blocks[s]), dim3(256), 0, st, (const double*)tmp, ws + p.part_off[s], planes, ho, w, wo, t, c1, c2);
        if (s + 1 < VM_SCALES) {
            const int h2 = p.H[s + 1], w2 = p.W[s + 1];
            double* xo = ws + p.plane_off[s + 1];
            double* yo = xo + (size_t)planes * h2 * w2;
            const unsigned gd = (unsigned)((planes * h2 * w2 + 255) / 256);
            if (s == 0) hipLaunchKernelGGL(vm_down_kernel<unsigned char>, dim3(gd), dim3(256), 0, st, x, y, xo, yo, planes, h, w, h2, w2);
            else hipLaunchKernelGGL(vm_down_kernel<double>, dim3(gd), dim3(256), 0, st, (const double*)(ws + p.plane_off[s]),
                                    (const double*)(ws + p.plane_off[s] + (size_t)planes * h * w), xo, yo, planes, h, w, h2, w2);
        }
        f.partial[s] = ws + p.part_off[s]; f.blocks[s] = p.blocks[s]; f.stride[s] = 2; f.pick[s] = s == VM_SCALES - 1 ? 0 : 1;
        f.count[s] = (double)planes * ho * wo;
    }
    hipLaunchKernelGGL(vm_sqerr_kernel, dim3(1024), dim3(256), 0, st, x, y, ws + p.part_off[VM_SCALES], planes * H * W);
    f.partial[VM_SCALES] = ws + p.part_off[VM_SCALES]; f.blocks[VM_SCALES] = 1024; f.stride[VM_SCALES] = 1; f.pick[VM_SCALES] = 0;
    f.count[VM_SCALES] = (double)planes * H * W;
    hipLaunchKernelGGL(vm_finish_kernel, dim3(1), dim3(256), 0, st, f, out6);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// the same for every image of a batch separately (val.py evaluates consecutive small images of one shape as one batch but reports
// per-image measures): image n's six values at out6n + 6 n, by the single-image sequence above launched N times on `stream` -- the same
// kernels on the same operands as N calls with N = 1 (bit-identical), one call across the ABI.  workspace: ic_val_metrics_workspace_bytes(1, C, H, W).
extern "C" int ic_val_metrics_per_image_u8_f64(const unsigned char* x, const unsigned char* y, int N, int C, int H, int W, double* out6n,
                                               void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(x && y && out6n && workspace && N > 0 && C > 0 && H > 0 && W > 0);
    const size_t img = (size_t)C * H * W;
    for (int n = 0; n < N; ++n) {
        const int rc = ic_val_metrics_u8_f64(x + n * img, y + n * img, 1, C, H, W, out6n + 6 * (size_t)n, workspace, workspace_bytes, stream);
        if (rc != IC_OK) return rc;
    }
    return IC_OK;
}

