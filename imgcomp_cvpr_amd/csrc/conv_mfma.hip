// Generic implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) for the strided
// layers that frame the residual stacks:
//   h2    5x5 / stride 2        64 -> 128   (reference code/autoencoder.py:223)
//   to_bn 5x5 / stride 2       128 -> C+1   (:237)
//   h12   5x5 / stride 2 transposed 128 -> 64 (:264)
// Every case is brought to ONE form: a stride-1 correlation over an output GRID with NTY x NTX taps,
//     out[co][gy][gx] = sum_{ty,tx,ci} Wp[ty][tx][ci][co] * in[ci][gy*PS + oy0 + ty][gx*PS + ox0 + tx]
//   * strided conv: PS = 2, taps = the 5x5 filter, oy0 = -pad_before (TF SAME: 1);
//   * transposed conv: split into its 4 output phases (py,px); phase outputs sit at (2gy+py, 2gx+px) and
//     read the input grid with PS = 1 through the filter taps of matching parity (3x3, 3x2, 2x3, 2x2
//     taps for k = 5) -- no zero insertion, no wasted MACs.  The tap <-> filter index map lives in the
//     packing kernel, so the main kernel never sees it.
// Same MFMA formulation as conv3x3_mfma.hip (A = filter fragments streamed from L2 into registers,
// B = halo tile staged through LDS, D[co][pixel]); here the filter fragments use a small register RING
// over the taps (prefetch distance RD-2 taps, slot reused two taps after its last MFMA), which keeps the
// register count low enough for 2-4 resident waves per SIMD: that occupancy, not a hand-built pipeline,
// hides the L2 and LDS latencies (measured on the 3x3 kernel: 3 work-groups per CU reach 95 % matrix-pipe
// occupancy in the main loop with the plain schedule).
// K order per output: chunk (8 ci) -> tap (ty,tx) -> k-step (2 ci); fixed and position independent.
#include "internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GKC 8

struct GArgs {
    const float* x; const float* wp; const float* scale; const float* shift; float* y;
    int N, IH, IW;          // input extent
    int GH, GW;             // output grid extent (per phase for transposed)
    int OH, OW, OS, py, px; // output tensor extent, grid->output stride and phase offset
    int oy0, ox0;           // input offset of tap (0,0) relative to grid position * PS
    int Cout, relu;
    int tiles_x, tiles_y;
};

// packed[((c*NT + t)*NCOT + n)*256 + l*4 + j] = W[ky(ty)][kx(tx)][ci = 8c + 2j + (l>>5)][co = 32n + (l&31)]
// (zero for co >= Cout).  ky = kyb + kys*ty, kx = kxb + kxs*tx.  transposed: filter is [kh][kw][cout][cin].
__global__ void pack_conv_mfma_kernel(const float* __restrict__ w, float* __restrict__ out, int KW, int Cin, int Cout,
                                      int transposed, int NTY, int NTX, int NCOT, int kyb, int kys, int kxb, int kxs,
                                      int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 3, l = (idx >> 2) & 63;
    int r = idx >> 8;
    const int n = r % NCOT; r /= NCOT;
    const int NT = NTY * NTX;
    const int t = r % NT, c = r / NT;
    const int ty = t / NTX, tx = t % NTX;
    const int ky = kyb + kys * ty, kx = kxb + kxs * tx;
    const int ci = GKC * c + 2 * j + (l >> 5), co = 32 * n + (l & 31);
    float v = 0.f;
    if (co < Cout)
        v = transposed ? w[(((size_t)ky * KW + kx) * Cout + co) * Cin + ci]
                       : w[(((size_t)ky * KW + kx) * Cin + ci) * Cout + co];
    out[idx] = v;
}

template <int NTY, int NTX, int PS, int CIN, int WM, int WN, int PT, int TR, int TC, int RD>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const GArgs a) {
    constexpr int NT = NTY * NTX;
    constexpr int NCH = CIN / GKC;
    constexpr int ROWS = (TR - 1) * PS + NTY, COLS = (TC - 1) * PS + NTX;
    constexpr int S = COLS, CS = ROWS * COLS, CHUNK = GKC * CS;
    constexpr int NST = (CHUNK + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per work-group");
    static_assert(TR * TC == 32 * PT * WN, "tile = PT accumulator tiles per wave x WN pixel groups");
    static_assert(NT % RD == 0 && RD >= 3, "ring depth must divide the tap count");
    static_assert(NST <= 32, "in-bounds mask is 32 bits");
    static_assert(CIN % GKC == 0, "Cin multiple of 8");
    __shared__ float lds[2][NST * 256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    int b = blockIdx.x;
    const int tx_ = b % a.tiles_x; b /= a.tiles_x;
    const int ty_ = b % a.tiles_y; const int n = b / a.tiles_y;
    const int cot = blockIdx.y * WM + wm;                 // this wave's 32-channel output tile
    const int ncot = gridDim.y * WM;
    const int gx0 = tx_ * TC, gy0 = ty_ * TR;
    const int IHW = a.IH * a.IW;
    const float* __restrict__ xin = a.x + (size_t)n * CIN * IHW;

    // staging plan
    int goff[NST];
    unsigned inb = 0;
    const int iy0 = gy0 * PS + a.oy0, ix0 = gx0 * PS + a.ox0;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / CS, rem = e - ci * CS;
        const int rr = rem / S, cc = rem - rr * S;
        const int iy = iy0 + rr, ix = ix0 + cc;
        const bool ok = (e < CHUNK) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        goff[i] = ok ? ci * IHW + iy * a.IW + ix : 0;
        inb |= (ok ? 1u : 0u) << i;
    }
    const int j = lane & 31, kh = lane >> 5;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * (PT * wn + p) + j;
        boff[p] = kh * CS + (q / TC) * PS * S + (q % TC) * PS;
    }
    // wp[(c*NT + t)*ncot*64] = A fragments of chunk c, tap t for this wave's output channels
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + (size_t)cot * 64 + lane;
    const int wstep = ncot * 64;

    f32x16 acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x4 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wp[(size_t)t * wstep];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const float v = xin[goff[i]];
        lds[0][tid + 256 * i] = ((inb >> i) & 1) ? v : 0.f;
    }
    __syncthreads();

    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < NCH;
        float st[NST];
        if (more) {
            const float* xc = xin + (size_t)(c + 1) * GKC * IHW;
#pragma unroll
            for (int i = 0; i < NST; ++i) st[i] = xc[goff[i]];
        }
        const float* __restrict__ L = lds[buf];
        const f32x4* wc = wp + (size_t)c * NT * wstep;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // ring: request tap g + RD - 2 into the slot whose last reader was tap g - 2
            {
                const int tn = t + RD - 2;
                if (tn < NT) ring[tn % RD] = wc[(size_t)tn * wstep];
                else if (more) ring[tn % RD] = wc[(size_t)tn * wstep];     // runs on into chunk c+1 (contiguous)
            }
            const int tapoff = (t / NTX) * S + (t % NTX);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float av = ring[t % RD][ks];
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const float bv = L[boff[p] + 2 * ks * CS + tapoff];
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p], 0, 0, 0);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NST; ++i) lds[buf ^ 1][tid + 256 * i] = ((inb >> i) & 1) ? st[i] : 0.f;
        }
        // raw barrier: own LDS traffic done, filter prefetch stays in flight (no vmcnt(0) drain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*kh channel, j pixel
    const float* __restrict__ scale = a.scale;
    const float* __restrict__ shift = a.shift;
    float* __restrict__ y = a.y;
    float sc[16], sh[16];
    bool cok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * cot + (r & 3) + 8 * (r >> 2) + 4 * kh;
        cok[r] = co < a.Cout;
        sc[r] = cok[r] ? scale[co] : 0.f;
        sh[r] = cok[r] ? shift[co] : 0.f;
    }
    const size_t OHW = (size_t)a.OH * a.OW;
    const size_t cbase = ((size_t)n * a.Cout + 32 * cot + 4 * kh) * OHW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * (PT * wn + p) + j;
        const int gy = gy0 + q / TC, gx = gx0 + q % TC;
        if (gy < a.GH && gx < a.GW) {
            const size_t pix = cbase + (size_t)(gy * a.OS + a.py) * a.OW + (gx * a.OS + a.px);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaf(acc[p][r], sc[r], sh[r]);
                if (a.relu) v = fmaxf(v, 0.f);
                if (cok[r]) y[pix + (size_t)((r & 3) + 8 * (r >> 2)) * OHW] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int ncot_for(int Cout) { return ic_cdiv(Cout, 32) <= 2 ? 2 : ic_cdiv(ic_cdiv(Cout, 32), 4) * 4; }

// number of taps of transposed-conv phase p (0/1) along one axis, and the offset of its first tap
static void phase_taps(int K, int pad, int p, int* nt, int* o0, int* kb) {
    // valid k: k == (p + pad) mod 2; input offset d(k) = (p + pad - k) / 2; taps ordered by ascending d
    const int k0 = (p + pad) & 1;
    const int cnt = (K - k0 + 1) / 2;
    const int kmax = k0 + 2 * (cnt - 1);
    *nt = cnt;
    *o0 = (p + pad - kmax) / 2;       // smallest input offset (exact: same parity)
    *kb = kmax;                        // tap 0 <-> k = kmax, tap t <-> k = kmax - 2t
}

static bool supported(int KH, int KW, int Cin, int Cout, int stride, int transposed) {
    if (KH != 5 || KW != 5 || stride != 2) return false;
    if (!transposed) return (Cin == 64 && Cout == 128) || (Cin == 128 && Cout <= 128);
    return Cin == 128 && Cout == 64;
}

extern "C" size_t ic_conv2d_mfma_packed_floats(int KH, int KW, int Cin, int Cout, int stride, int transposed) {
    if (!supported(KH, KW, Cin, Cout, stride, transposed)) return 0;
    // strided conv: one block of all taps; transposed conv: its four phases back to back (same total taps)
    return (size_t)(Cin / GKC) * KH * KW * ncot_for(Cout) * 256;
}

extern "C" int ic_pack_conv2d_mfma_f32(const float* w_tf, float* w_packed, int KH, int KW, int Cin, int Cout,
                                       int stride, int transposed, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    if (!supported(KH, KW, Cin, Cout, stride, transposed)) return IC_ERR_UNSUPPORTED;
    const int ncot = ncot_for(Cout), nch = Cin / GKC;
    hipStream_t st = (hipStream_t)stream;
    if (!transposed) {
        const int total = nch * KH * KW * ncot * 256;
        hipLaunchKernelGGL(pack_conv_mfma_kernel, dim3(ic_cdiv(total, 256)), dim3(256), 0, st, w_tf, w_packed, KW, Cin,
                           Cout, 0, KH, KW, ncot, 0, 1, 0, 1, total);
    } else {
        const int pad = 1;   // SAME forward conv 2H -> H with k = 5: pad_before 1
        size_t off = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                int nty, ntx, oy0, ox0, kyb, kxb;
                phase_taps(KH, pad, py, &nty, &oy0, &kyb);
                phase_taps(KW, pad, px, &ntx, &ox0, &kxb);
                const int total = nch * nty * ntx * ncot * 256;
                hipLaunchKernelGGL(pack_conv_mfma_kernel, dim3(ic_cdiv(total, 256)), dim3(256), 0, st, w_tf,
                                   w_packed + off, KW, Cin, Cout, 1, nty, ntx, ncot, kyb, -2, kxb, -2, total);
                off += (size_t)total;
            }
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}

#define G_LAUNCH(NTY_, NTX_, PS_, CIN_, WM_, WN_, PT_, TR_, TC_, RD_)                                         \
    do {                                                                                                       \
        a.tiles_x = ic_cdiv(a.GW, TC_); a.tiles_y = ic_cdiv(a.GH, TR_);                                        \
        hipLaunchKernelGGL((conv_mfma_kernel<NTY_, NTX_, PS_, CIN_, WM_, WN_, PT_, TR_, TC_, RD_>),            \
                           dim3(a.tiles_x * a.tiles_y * a.N, ncot / WM_), dim3(256), 0, st, a);                \
    } while (0)

extern "C" int ic_conv2d_mfma_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                         float* y, int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                                         int transposed, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y && N > 0 && H > 0 && W > 0);
    if (!supported(KH, KW, Cin, Cout, stride, transposed)) return IC_ERR_UNSUPPORTED;
    if ((long long)Cin * H * W >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int ncot = ncot_for(Cout);
    GArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.y = y;
    a.N = N; a.IH = H; a.IW = W; a.Cout = Cout; a.relu = relu;
    if (!transposed) {
        a.wp = w_packed;
        a.GH = ic_cdiv(H, 2); a.GW = ic_cdiv(W, 2); a.OH = a.GH; a.OW = a.GW; a.OS = 1; a.py = 0; a.px = 0;
        a.oy0 = -ic_same_pad_before(H, KH, 2); a.ox0 = -ic_same_pad_before(W, KW, 2);
        if (Cin == 64) G_LAUNCH(5, 5, 2, 64, 4, 1, 1, 2, 16, 5);          // h2: 4 co tiles x 32 px
        else G_LAUNCH(5, 5, 2, 128, 2, 2, 1, 4, 16, 5);                   // to_bn: 2 co tiles x 64 px
    } else {
        a.GH = H; a.GW = W; a.OH = 2 * H; a.OW = 2 * W; a.OS = 2;
        const int pad = 1, nch = Cin / GKC;
        size_t off = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                int nty, ntx, kyb, kxb;
                phase_taps(KH, pad, py, &nty, &a.oy0, &kyb);
                phase_taps(KW, pad, px, &ntx, &a.ox0, &kxb);
                a.wp = w_packed + off; a.py = py; a.px = px;
                // h12: 2 co tiles x 2 pixel groups of 32 px
                if (nty == 3 && ntx == 3) G_LAUNCH(3, 3, 1, 128, 2, 2, 1, 4, 16, 3);
                else if (nty == 3 && ntx == 2) G_LAUNCH(3, 2, 1, 128, 2, 2, 1, 4, 16, 3);
                else if (nty == 2 && ntx == 3) G_LAUNCH(2, 3, 1, 128, 2, 2, 1, 4, 16, 3);
                else G_LAUNCH(2, 2, 1, 128, 2, 2, 1, 4, 16, 4);
                off += (size_t)nch * nty * ntx * ncot * 256;
            }
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}
