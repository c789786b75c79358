// The image-side edge layer of the decoder: h13 = 5x5 / stride-2 transposed conv 64 -> 3 + BN + de-normalise +
// clip (reference code/autoencoder.py:265-267).  3 output channels are no matrix-core shape (3/32 of a tile);
// the layer is bound by streaming its 64-channel input once (256 B per output pixel) and writing 12 B.
// One lane = one INPUT-grid position = the 2x2 output pixels x Cout it feeds (12 accumulators): per input channel
// it loads the 3x3 input neighbourhood once (9 loads) and applies all 25 taps x Cout (75 FMAs) -- 4x fewer lanes
// and 2.8x more FMAs per load than the phase-per-lane form of conv_direct.hip.  The filter is re-laid into LDS as
// [ci][ky][kx][4] once per work-group and read back as broadcast 16-byte words.
// fp32 FMA chain per output in (ci, dy, dx) order.
#include "internal.h"

#define DE_TX 32
#define DE_TY 8

template <int CO>
__global__ __launch_bounds__(256) void deconv5_small_cout_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [Cin][25][4]
    const int tid = threadIdx.x;
    const int n = blockIdx.z;
    // filter: TF conv2d_transpose layout [kh][kw][cout][cin]
    for (int e = tid; e < a.Cin * 25 * 4; e += 256) {
        const int co = e & 3, tap = (e >> 2) % 25, ci = e / 100;
        wl[e] = co < a.Cout ? a.w[((size_t)tap * a.Cout + co) * a.Cin + ci] : 0.f;
    }
    __syncthreads();
    const int qx = blockIdx.x * DE_TX + (tid & (DE_TX - 1));
    const int qy = blockIdx.y * DE_TY + tid / DE_TX;
    const bool live = qx < a.W && qy < a.H;
    const int cx = min(qx, a.W - 1), cy = min(qy, a.H - 1);
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * a.Cin * HW;
    // neighbour offsets and validity (zero outside the input)
    int noff[9];
    bool nok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = cy + k / 3 - 1, ix = cx + k % 3 - 1;
        nok[k] = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        noff[k] = nok[k] ? iy * a.W + ix : 0;
    }
    float acc[2][2][CO];
#pragma unroll
    for (int i = 0; i < 4 * CO; ++i) (&acc[0][0][0])[i] = 0.f;

    for (int ci = 0; ci < a.Cin; ++ci) {
        const float* __restrict__ xp = xin + (size_t)ci * HW;
        float nb[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { const float v = xp[noff[k]]; nb[k] = nok[k] ? v : 0.f; }
        const float4* __restrict__ wc = reinterpret_cast<const float4*>(wl) + ci * 25;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int dy = k / 3 - 1, dx = k % 3 - 1;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int ky = py + 1 - 2 * dy;           // SAME pads of the 2H -> H forward conv, k = 5: 1
                if (ky < 0 || ky > 4) continue;
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int kx = px + 1 - 2 * dx;
                    if (kx < 0 || kx > 4) continue;
                    const float4 w4 = wc[ky * 5 + kx];
                    acc[py][px][0] = fmaf(nb[k], w4.x, acc[py][px][0]);
                    if (CO > 1) acc[py][px][1] = fmaf(nb[k], w4.y, acc[py][px][1]);
                    if (CO > 2) acc[py][px][2] = fmaf(nb[k], w4.z, acc[py][px][2]);
                    if (CO > 3) acc[py][px][3] = fmaf(nb[k], w4.w, acc[py][px][3]);
                }
            }
        }
    }
    if (!live) return;
    const size_t ohw = (size_t)a.OH * a.OW;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        if (co >= a.Cout) break;
        const float sc = a.scale[co], sh = a.shift[co];
        float om = 0.f, os = 1.f;
        const bool dn = a.out_mean != nullptr || (a.builtin_norm & 2);
        if (a.out_mean) { om = a.out_mean[co]; os = a.out_std[co]; }
        else if (a.builtin_norm & 2) { om = IC_IMG_MEAN[co]; os = IC_IMG_STD[co]; }
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float2 o;
            float* op = &o.x;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                float v = fmaf(acc[py][px][co], sc, sh);
                if (a.relu) v = fmaxf(v, 0.f);
                if (dn) v = fminf(fmaxf(__fadd_rn(__fmul_rn(v, os), om), 0.f), 255.f);
                else if (a.builtin_norm & 4) v = fminf(fmaxf(v, 0.f), 255.f);
                op[px] = v;
            }
            float* dst = a.y + ((size_t)n * a.Cout + co) * ohw + (size_t)(2 * qy + py) * a.OW + 2 * qx;
            *reinterpret_cast<float2*>(dst) = o;
        }
    }
}

// returns IC_ERR_UNSUPPORTED when the shape is not this kernel's (caller falls back to the generic one)
int icx_deconv5_small_cout(const ConvArgs& a, hipStream_t st) {
    if (a.KH != 5 || a.KW != 5 || a.Cout > 4 || a.res1 || a.res2) return IC_ERR_UNSUPPORTED;
    const size_t lds = (size_t)a.Cin * 25 * 4 * sizeof(float);
    if (lds > 64 * 1024) return IC_ERR_UNSUPPORTED;
    dim3 g(ic_cdiv(a.W, DE_TX), ic_cdiv(a.H, DE_TY), a.N);
    hipLaunchKernelGGL((deconv5_small_cout_kernel<4>), g, dim3(256), lds, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
