// Generic direct convolution / stride-2 transposed convolution with fused BN + activation epilogue.
//
// Covers the six "odd" conv sites of the CVPR autoencoder (reference code/autoencoder.py:222 h1,
// :223 h2, :237 to_bn, :251 from_bn, :264 h12, :265 h13) and serves as the any-shape fallback for
// the 3x3 residual convs.  One lane = one output pixel x COB output channels: the input value is a
// per-lane (coalesced along W) load, the filter taps are wave-uniform and come through the scalar
// cache, so every vector load feeds COB FMAs.  fp32 FMA chain in (ci, ky, kx) order per output.
#include "internal.h"


template <int COB, bool TRANSPOSED, bool CONTIG = false>
__global__ __launch_bounds__(256) void conv2d_direct_kernel(const ConvArgs a) {
    const int n = blockIdx.z;
    int cob, py = 0, px = 0;
    if (TRANSPOSED) { cob = blockIdx.y >> 2; py = (blockIdx.y >> 1) & 1; px = blockIdx.y & 1; }
    else cob = blockIdx.y;
    const int co0 = cob * COB;
    // pixel grid this lane walks: output pixels (conv) or input-grid positions of one phase (deconv)
    const int GW = TRANSPOSED ? a.W : a.OW, GH = TRANSPOSED ? a.H : a.OH;
    int p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < GH * GW;
    if (!live) p = GH * GW - 1;
    const int gy = p / GW, gx = p - gy * GW;

    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;

    int wofs[COB];   // clamped channel offsets (wave-uniform)
#pragma unroll
    for (int j = 0; j < COB; ++j) wofs[j] = min(co0 + j, a.Cout - 1) * a.w_sco;

    const int HW = a.H * a.W;
    const float* xn = a.x + (size_t)n * a.Cin * HW;
    const int tapstride = a.Cin * a.Cout;
    const int ky0 = TRANSPOSED ? ((py + a.pt) & 1) : 0, kx0 = TRANSPOSED ? ((px + a.pl) & 1) : 0;
    const int kstep = TRANSPOSED ? 2 : 1;

    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* xp = xn + (size_t)ci * HW;
        float m = 0.f, s = 1.f;
        const bool norm_in = a.in_mean != nullptr || (a.builtin_norm & 1);
        if (a.in_mean) { m = a.in_mean[ci]; s = a.in_std[ci]; }
        else if (a.builtin_norm & 1) { m = IC_IMG_MEAN[ci]; s = IC_IMG_STD[ci]; }
        for (int ky = ky0; ky < a.KH; ky += kstep) {
            const int iy = TRANSPOSED ? gy + ((py + a.pt - ky) >> 1) : gy * a.stride + ky - a.pt;
            const bool vy = iy >= 0 && iy < a.H;
            for (int kx = kx0; kx < a.KW; kx += kstep) {
                const int ix = TRANSPOSED ? gx + ((px + a.pl - kx) >> 1) : gx * a.stride + kx - a.pl;
                const bool v = vy && ix >= 0 && ix < a.W;
                float xv = 0.f;
                if (v) {
                    xv = xp[iy * a.W + ix];
                    if (norm_in) xv = (xv - m) / s;
                }
                const float* wp = a.w + (size_t)(ky * a.KW + kx) * tapstride + (size_t)ci * a.w_sci;
#pragma unroll
                for (int j = 0; j < COB; ++j) acc[j] = fmaf(xv, CONTIG ? wp[co0 + j] : wp[wofs[j]], acc[j]);
            }
        }
    }
    if (!live) return;
    const int oy = TRANSPOSED ? 2 * gy + py : gy, ox = TRANSPOSED ? 2 * gx + px : gx;
    const size_t ohw = (size_t)a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < COB; ++j) {
        const int co = co0 + j;
        if (co < a.Cout) {
            float v = fmaf(acc[j], a.scale[co], a.shift[co]);
            if (a.relu) v = fmaxf(v, 0.f);
            const size_t o = ((size_t)n * a.Cout + co) * ohw + (size_t)oy * a.OW + ox;
            if (a.res1) v += a.res1[o];
            if (a.res2) v += a.res2[o];
            if (a.out_mean) v = fminf(fmaxf(__fadd_rn(__fmul_rn(v, a.out_std[co]), a.out_mean[co]), 0.f), 255.f);
            else if (a.builtin_norm & 2) v = fminf(fmaxf(__fadd_rn(__fmul_rn(v, IC_IMG_STD[co]), IC_IMG_MEAN[co]), 0.f), 255.f);
            else if (a.builtin_norm & 4) v = fminf(fmaxf(v, 0.f), 255.f);
            a.y[o] = v;
        }
    }
}

template <bool TRANSPOSED>
static int launch_direct(const ConvArgs& a, hipStream_t st) {
    const int GH = TRANSPOSED ? a.H : a.OH, GW = TRANSPOSED ? a.W : a.OW;
    const int pb = ic_cdiv(GH * GW, 256);
    // small Cout (h13: 3) keeps registers low; otherwise 16 accumulators per lane
    if (!TRANSPOSED && a.Cin <= 4 && a.Cout % 32 == 0 && a.w_sco == 1) {
        // h1: few input channels, many outputs: 32 accumulators per lane, filter taps as runs of 32 scalars
        dim3 g(pb, a.Cout / 32, a.N);
        hipLaunchKernelGGL((conv2d_direct_kernel<32, false, true>), g, dim3(256), 0, st, a);
    } else if (a.Cout <= 4) {
        dim3 g(pb, (TRANSPOSED ? 4 : 1) * ic_cdiv(a.Cout, 4), a.N);
        hipLaunchKernelGGL((conv2d_direct_kernel<4, TRANSPOSED>), g, dim3(256), 0, st, a);
    } else if (a.Cout % 16 != 0 && a.Cout % 11 == 0) {   // to_bn: 33 = 3 x 11
        dim3 g(pb, (TRANSPOSED ? 4 : 1) * ic_cdiv(a.Cout, 11), a.N);
        hipLaunchKernelGGL((conv2d_direct_kernel<11, TRANSPOSED>), g, dim3(256), 0, st, a);
    } else {
        dim3 g(pb, (TRANSPOSED ? 4 : 1) * ic_cdiv(a.Cout, 16), a.N);
        hipLaunchKernelGGL((conv2d_direct_kernel<16, TRANSPOSED>), g, dim3(256), 0, st, a);
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}

int icx_conv2d(ConvArgs a, bool transposed, hipStream_t st) {
    if (!transposed) {
        a.OH = ic_cdiv(a.H, a.stride); a.OW = ic_cdiv(a.W, a.stride);
        a.pt = ic_same_pad_before(a.H, a.KH, a.stride); a.pl = ic_same_pad_before(a.W, a.KW, a.stride);
        a.w_sci = a.Cout; a.w_sco = 1;                 // TF conv2d filter [kh,kw,cin,cout]
        if (a.Cin == 3 && a.Cout == 64 && a.KH == 5 && a.KW == 5 && a.stride == 2) {   // h1 on the matrix cores
            const int rc = icx_conv5s2_cin3_mfma(a, st);
            if (rc != IC_ERR_UNSUPPORTED) return rc;
        }
        if (a.out_phases) return IC_ERR_UNSUPPORTED;    // only the h1 kernel writes phase planes
        return launch_direct<false>(a, st);
    }
    a.stride = 2; a.OH = 2 * a.H; a.OW = 2 * a.W;
    // pads of the SAME forward conv (2H -> H) whose adjoint this is (k=3: 0, k=5: 1)
    a.pt = ic_same_pad_before(2 * a.H, a.KH, 2); a.pl = ic_same_pad_before(2 * a.W, a.KW, 2);
    a.w_sci = 1; a.w_sco = a.Cin;                      // TF conv2d_transpose filter [kh,kw,cout,cin]
    if (a.Cout <= 4 && a.KH == 5 && a.KW == 5) {
        int rc = icx_deconv5_cout3_mfma(a, st);          // h13: 64 input channels, matrix cores
        if (rc != IC_ERR_UNSUPPORTED) return rc;
        rc = icx_deconv5_small_cout(a, st);
        if (rc != IC_ERR_UNSUPPORTED) return rc;
    }
    if (a.KH == 3 && a.KW == 3 && a.Cout == 128) {     // from_bn on the matrix cores
        const int rc = icx_deconv3_mfma(a, st);
        if (rc != IC_ERR_UNSUPPORTED) return rc;
    }
    return launch_direct<true>(a, st);
}

extern "C" int ic_conv2d_bn_act_f32(const float* x, const float* w, const float* scale, const float* shift,
                                    const float* res1, const float* res2, float* y,
                                    int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int relu,
                                    const float* in_mean, const float* in_std, ic_stream_t stream) {
    IC_CHECK_ARG(x && w && scale && shift && y);
    IC_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0);
    IC_CHECK_ARG((in_mean == nullptr) == (in_std == nullptr));
    if (stride != 1 && stride != 2) return IC_ERR_UNSUPPORTED;
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.y = y;
    a.in_mean = in_mean; a.in_std = in_std;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KH = KH; a.KW = KW; a.stride = stride; a.relu = relu;
    return icx_conv2d(a, false, (hipStream_t)stream);
}

extern "C" int ic_deconv2d_bn_act_f32(const float* x, const float* w, const float* scale, const float* shift,
                                      float* y, int N, int Cin, int H, int W, int Cout, int KH, int KW, int relu,
                                      const float* out_mean, const float* out_std, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && w && scale && shift && y);
    IC_CHECK_ARG(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0);
    IC_CHECK_ARG((out_mean == nullptr) == (out_std == nullptr));
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.out_mean = out_mean; a.out_std = out_std;
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KH = KH; a.KW = KW; a.relu = relu;
    a.tune = flags & 0xff;                                   // IC_EDGE_TILES_PER_WG(n)
    return icx_conv2d(a, true, (hipStream_t)stream);
}
