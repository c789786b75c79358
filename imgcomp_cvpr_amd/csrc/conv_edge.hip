// The image-side edge layer of the decoder: h13 = 5x5 / stride-2 transposed conv 64 -> 3 + BN + de-normalise +
// clip (reference code/autoencoder.py:265-267).  3 output channels are no matrix-core shape (3/32 of a tile);
// the layer is bound by streaming its 64-channel input once (256 B per output pixel) and writing 12 B.
// One lane = one INPUT-grid position = the 2x2 output pixels x Cout it feeds (12 accumulators): per input channel
// it loads the 3x3 input neighbourhood once (9 loads) and applies all 25 taps x Cout (75 FMAs) -- 4x fewer lanes
// and 2.8x more FMAs per load than the phase-per-lane form of conv_direct.hip.  The filter is re-laid into LDS as
// [ci][ky][kx][4] once per work-group and read back as broadcast 16-byte words.
// fp32 FMA chain per output in (ci, dy, dx) order.
#include "internal.h"
#include <algorithm>

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

// ------------------------------------------------------------------------------------------------
// from_bn: 3x3 / stride-2 transposed conv C -> 128 + BN + ReLU (reference code/autoencoder.py:251) on the fp32
// matrix cores.  The SAME pads of the adjoint forward conv (2H -> H, k = 3) are 0, so output (2gy+py, 2gx+px)
// reads the input grid at (gy - dy, gx - dx) through the taps ky = py + 2dy, kx = px + 2dx (< 3): the four phases
// use 4, 2, 2 and 1 taps and share the four shifted views of ONE staged input tile.
//   * work-group = 2x16 grid positions x all 128 output channels; wave w owns channels [32w, 32w+32) for all four
//     phases (4 + 1 accumulator tiles: the 4-tap phase keeps two so that no MFMA waits on its predecessor);
//   * the whole K = 9 x Cin fits: the input tile (Cin x 3 x 17, zero outside the image) is staged once in LDS, the
//     filter fragments are read straight from the TF layout [ky][kx][cout][cin] -- lane (co, kh) takes the
//     contiguous run cin = 32c + 16kh .. +15 of each tap, i.e. MFMA k-step s of chunk c multiplies input channels
//     (32c + s, 32c + 16 + s); any pairing is valid as long as the B operand follows it;
//   * per output the sum runs chunk -> s -> taps, fixed and position independent.
// ------------------------------------------------------------------------------------------------
typedef float e_f32x16 __attribute__((ext_vector_type(16)));
typedef float e_f32x4 __attribute__((ext_vector_type(4)));

#define D3_TR 2
#define D3_TC 16
#define D3_S (D3_TC + 1)
#define D3_CS ((D3_TR + 1) * D3_S)

template <int CIN>
__global__ __launch_bounds__(256) void deconv3_mfma_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    static_assert(CIN % 32 == 0, "input channels in chunks of 32");
    constexpr int NE = CIN * D3_CS, NST = (NE + 255) / 256, NCH = CIN / 32;
    __shared__ float lds[NST * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int gx0 = tx * D3_TC, gy0 = ty * D3_TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * CIN * HW;

    // stage the tile with its top / left halo
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / D3_CS, rem = e - ci * D3_CS;
        const int rr = rem / D3_S, cc = rem - rr * D3_S;
        const int iy = gy0 - 1 + rr, ix = gx0 - 1 + cc;
        const bool ok = e < NE && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const float v = xin[ok ? ci * HW + iy * a.W + ix : 0];
        lds[e] = ok ? v : 0.f;
    }

    const int j = lane & 31, kh = lane >> 5;
    // filter fragments: [tap][chunk] runs of 16 input channels
    const float* __restrict__ wl = a.w + (size_t)(32 * wave + j) * CIN + 16 * kh;
    const float* __restrict__ L = lds + (16 * kh) * D3_CS + (1 + j / D3_TC) * D3_S + 1 + (j % D3_TC);

    e_f32x16 p00a, p00b, p01, p10, p11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { p00a[r] = 0.f; p00b[r] = 0.f; p01[r] = 0.f; p10[r] = 0.f; p11[r] = 0.f; }

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        e_f32x4 A[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                A[t][q] = *reinterpret_cast<const e_f32x4*>(wl + (size_t)t * 128 * CIN + 32 * c + 4 * q);
        if (c == 0) __syncthreads();
        // the four shifted views of step s + 1 are read behind the nine MFMAs of step s
        float bv[2][4];
        auto views = [&](int s, int buf) __attribute__((always_inline)) {
            const float* Ls = L + (32 * c + s) * D3_CS;
            bv[buf][0] = Ls[0]; bv[buf][1] = Ls[-1]; bv[buf][2] = Ls[-D3_S]; bv[buf][3] = Ls[-D3_S - 1];
        };
        views(0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) views(s + 1, (s + 1) & 1);
            const float b00 = bv[s & 1][0], b0m = bv[s & 1][1], bm0 = bv[s & 1][2], bmm = bv[s & 1][3];
#define D3_A(t) A[t][s >> 2][s & 3]
            p00a = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(0), b00, p00a, 0, 0, 0);   // ky 0, kx 0
            p01 = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(1), b00, p01, 0, 0, 0);     // ky 0, kx 1
            p10 = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(3), b00, p10, 0, 0, 0);     // ky 1, kx 0
            p11 = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(4), b00, p11, 0, 0, 0);     // ky 1, kx 1
            p00b = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(2), b0m, p00b, 0, 0, 0);   // ky 0, kx 2: gx - 1
            p10 = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(5), b0m, p10, 0, 0, 0);     // ky 1, kx 2
            p00a = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(6), bm0, p00a, 0, 0, 0);   // ky 2, kx 0: gy - 1
            p01 = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(7), bm0, p01, 0, 0, 0);     // ky 2, kx 1
            p00b = __builtin_amdgcn_mfma_f32_32x32x2f32(D3_A(8), bmm, p00b, 0, 0, 0);   // ky 2, kx 2
#undef D3_A
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 4 && s + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*kh channel of the wave's tile, j = grid position
    const int gy = gy0 + j / D3_TC, gx = gx0 + j % D3_TC;
    if (gy >= a.H || gx >= a.W) return;
    const size_t ohw = (size_t)a.OH * a.OW;
    float* __restrict__ yb = a.y + ((size_t)n * a.Cout + 32 * wave + 4 * kh) * ohw + (size_t)(2 * gy) * a.OW + 2 * gx;
    const float* __restrict__ scp = a.scale + 32 * wave + 4 * kh;
    const float* __restrict__ shp = a.shift + 32 * wave + 4 * kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int crow = (r & 3) + 8 * (r >> 2);
        const float sc = scp[crow], sh = shp[crow];
        float2 o0, o1;
        o0.x = fmaf(p00a[r] + p00b[r], sc, sh); o0.y = fmaf(p01[r], sc, sh);
        o1.x = fmaf(p10[r], sc, sh); o1.y = fmaf(p11[r], sc, sh);
        if (a.relu) {
            o0.x = fmaxf(o0.x, 0.f); o0.y = fmaxf(o0.y, 0.f); o1.x = fmaxf(o1.x, 0.f); o1.y = fmaxf(o1.y, 0.f);
        }
        float* dst = yb + (size_t)crow * ohw;
        *reinterpret_cast<float2*>(dst) = o0;
        *reinterpret_cast<float2*>(dst + a.OW) = o1;
    }
}

int icx_deconv3_mfma(const ConvArgs& a, hipStream_t st) {
    if (a.KH != 3 || a.KW != 3 || a.Cout != 128 || a.pt != 0 || a.pl != 0 || a.res1 || a.res2 || a.in_mean ||
        a.out_mean || a.builtin_norm || a.w_sci != 1 || a.w_sco != a.Cin)
        return IC_ERR_UNSUPPORTED;
    if ((long long)a.Cin * a.H * a.W >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    const int tiles_x = ic_cdiv(a.W, D3_TC), tiles_y = ic_cdiv(a.H, D3_TR);
    const dim3 g((unsigned)(tiles_x * tiles_y * a.N));
    if (a.Cin == 32) hipLaunchKernelGGL((deconv3_mfma_kernel<32>), g, dim3(256), 0, st, a, tiles_x, tiles_y);
    else if (a.Cin == 64) hipLaunchKernelGGL((deconv3_mfma_kernel<64>), g, dim3(256), 0, st, a, tiles_x, tiles_y);
    else return IC_ERR_UNSUPPORTED;
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ------------------------------------------------------------------------------------------------
// h1: 5x5 / stride-2 conv 3 -> 64 + BN + ReLU with the input normalisation folded into the load (reference
// code/autoencoder.py:222, :136-150) on the fp32 matrix cores.  K = 3 x 25 is laid out as (ci, ky pair, kx): the two
// k rows of one v_mfma_f32_32x32x2_f32 are filter rows ky = 2p and 2p + 1 of the same (ci, kx), so the B operand of
// lane half kh is one LDS read at a compile-time offset from a per-lane base that already contains kh rows -- no
// index arithmetic in the loop.  ky = 5 does not exist: its filter fragment is zero and the staged tile carries one
// extra (finite) row for it, 45 k-steps instead of 37.5.
//   * work-group = 4 output rows x 32 columns x 64 channels; wave w = row w, two channel tiles (two accumulators);
//   * the normalised input tile (3 x 12 x 67, zero outside the image = TF SAME padding of the normalised image) is
//     staged once with even and odd columns split, so the stride-2 reads of 32 lanes are bank-conflict free;
//   * the filter is copied to LDS in its TF layout [ky][kx][ci][co]; a lane's fragments are runs of 32 channels.
// Per output the sum runs ci -> ky pair -> kx, fixed and position independent.
// ------------------------------------------------------------------------------------------------
#define H1_TR 4
#define H1_TC 32
#define H1_ROWS ((H1_TR - 1) * 2 + 6)          // 5 filter rows + the phantom sixth
#define H1_COLS ((H1_TC - 1) * 2 + 5)
#define H1_HALF ((H1_COLS + 1) / 2)
#define H1_S (2 * H1_HALF)
#define H1_CS (H1_ROWS * H1_S)

__global__ __launch_bounds__(256) void conv5s2_cin3_mfma_kernel(const ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NE = 3 * H1_ROWS * H1_COLS, NST = (NE + 255) / 256;
    __shared__ float lds[3 * H1_CS];
    __shared__ __attribute__((aligned(16))) float wl[25 * 3 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int ox0 = tx * H1_TC, oy0 = ty * H1_TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * 3 * HW;
    const int iy0 = 2 * oy0 - a.pt, ix0 = 2 * ox0 - a.pl;
    const bool norm_in = a.in_mean != nullptr || (a.builtin_norm & 1);

#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / (H1_ROWS * H1_COLS), rem = e - ci * (H1_ROWS * H1_COLS);
        const int rr = rem / H1_COLS, cc = rem - rr * H1_COLS;
        const int iy = iy0 + rr, ix = ix0 + cc;
        const bool live = e < NE;
        const bool ok = live && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        float v = xin[ok ? ci * HW + iy * a.W + ix : 0];
        if (norm_in) {
            const int cm = live ? ci : 0;
            const float m = a.in_mean ? a.in_mean[cm] : IC_IMG_MEAN[cm];
            const float s = a.in_mean ? a.in_std[cm] : IC_IMG_STD[cm];
            v = (v - m) / s;
        }
        if (live) lds[ci * H1_CS + rr * H1_S + (cc & 1) * H1_HALF + (cc >> 1)] = ok ? v : 0.f;
    }

    // the filter goes through LDS as it lies in memory (19 KB, 16-byte moves): 90 scalar loads per lane straight from
    // the TF layout would keep the texture-address unit busier than the matrix pipes
    for (int e = tid; e < 25 * 3 * 64 / 4; e += 256)
        reinterpret_cast<float4*>(wl)[e] = reinterpret_cast<const float4*>(a.w)[e];
    __syncthreads();

    const int j = lane & 31, kh = lane >> 5;
    const float* __restrict__ L = lds + (2 * wave + kh) * H1_S + j;
    e_f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // step s = (ci*3 + p)*5 + kx multiplies filter row ky = 2p + kh; the operands of step s + 1 (two filter fragments,
    // one input value, all from LDS) are read behind the MFMAs of step s
    float fa[2][2], fb[2];
    auto operands = [&](int s, int buf) __attribute__((always_inline)) {
        const int ci = s / 15, p = (s / 5) % 3, kx = s % 5;
        const int ky = 2 * p + kh;
        const float* wp = wl + (((p < 2 ? ky : 4) * 5 + kx) * 3 + ci) * 64 + j;
        const float w0 = wp[0], w1 = wp[32];
        fa[buf][0] = (p < 2 || kh == 0) ? w0 : 0.f;            // ky = 5 does not exist
        fa[buf][1] = (p < 2 || kh == 0) ? w1 : 0.f;
        fb[buf] = L[ci * H1_CS + 2 * p * H1_S + (kx & 1) * H1_HALF + (kx >> 1)];
    };
    operands(0, 0);
#pragma unroll
    for (int s = 0; s < 45; ++s) {
        if (s + 1 < 45) operands(s + 1, (s + 1) & 1);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][0], fb[s & 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][1], fb[s & 1], acc1, 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);

    const int gy = oy0 + wave, gx = ox0 + j;
    if (gy >= a.OH || gx >= a.OW) return;
    const size_t ohw = (size_t)a.OH * a.OW;
    // plain [n][co][gy][gx], or the four phases of the map as planes [n][2 py + px][co][gy / 2][gx / 2] (a.out_phases)
    const size_t cs = a.out_phases ? ohw / 4 : ohw;
    const size_t o0 = a.out_phases ? (((size_t)n * 4 + 2 * (gy & 1) + (gx & 1)) * 64 + 4 * kh) * cs + (size_t)(gy >> 1) * (a.OW >> 1) + (gx >> 1)
                                   : ((size_t)n * 64 + 4 * kh) * ohw + (size_t)gy * a.OW + gx;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * t + (r & 3) + 8 * (r >> 2);          // + 4 kh in the pointers
            float v = fmaf(t ? acc1[r] : acc0[r], a.scale[co + 4 * kh], a.shift[co + 4 * kh]);
            if (a.relu) v = fmaxf(v, 0.f);
            const size_t o = o0 + (size_t)co * cs;
            if (a.res1) v += a.res1[o];
            if (a.res2) v += a.res2[o];
            a.y[o] = v;
        }
}

int icx_conv5s2_cin3_mfma(const ConvArgs& a, hipStream_t st) {
    if (a.KH != 5 || a.KW != 5 || a.stride != 2 || a.Cin != 3 || a.Cout != 64 || a.w_sco != 1 || a.w_sci != 64 ||
        a.out_mean || (a.builtin_norm & ~1))
        return IC_ERR_UNSUPPORTED;
    if ((long long)a.H * a.W * 3 >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    if (a.out_phases && ((a.OH | a.OW) & 1 || a.res1 || a.res2)) return IC_ERR_UNSUPPORTED;
    const int tiles_x = ic_cdiv(a.OW, H1_TC), tiles_y = ic_cdiv(a.OH, H1_TR);
    hipLaunchKernelGGL(conv5s2_cin3_mfma_kernel, dim3((unsigned)(tiles_x * tiles_y * a.N)), dim3(256), 0, st, a, tiles_x,
                       tiles_y);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ------------------------------------------------------------------------------------------------
// h13 on the matrix cores: v_mfma_f32_16x16x4_f32 with M = (output channel, output phase) -- 3 x 4 = 12 of 16 rows --
// N = 16 input-grid positions, K = (input channel, 3x3 input neighbourhood) = 64 x 9.  Row (co, py, px) of the A
// operand holds filter tap ky = py + 1 - 2dy, kx = px + 1 - 2dx of neighbour (dy, dx), zero where that tap does
// not exist (25 of the 36 phase x neighbour pairs do), so ONE k-loop produces the 2x2 output pixels x Cout of every
// grid position: 52 % of the issued MACs are useful, against < 10 % for any tiling that puts Cout alone on M.
//   * work-group = a run of x-adjacent tiles of 8 x 16 grid positions; wave w = grid rows 2w, 2w+1 of each tile (two N
//     tiles, two accumulators each so that consecutive MFMAs never chain);
//   * the filter fragments (16 k-steps x 9 neighbours x 64 lanes, 36 KB) are gathered once per work-group from the TF
//     layout [ky][kx][cout][cin] into LDS in operand order.  Lane group kg of the MFMA takes the contiguous channels
//     16kg .. 16kg+15 (any assignment of channels to k rows is valid as long as A and B agree), so the gather is nine
//     16-byte loads per lane;
//   * the input streams through a double buffer in chunks of 16 channels (4 k-steps), each a 10 x 24 window per
//     channel moved as aligned float4 (dword loads would keep the texture-address unit as busy as the matrix pipes);
//     requests run two chunks ahead of the MFMAs and continue across the tiles of the run;
//   * the LDS operands of k-step c+1 are read before the MFMAs of step c are issued;
//   * D puts rows 4q..4q+3 on lane group q: lane (q, j) ends up with the 2x2 output block of channel q at grid
//     position j and writes it as two float2.
// Per output the sum runs k-step (channels c, 16+c, 32+c, 48+c) -> neighbour, even and odd neighbours in separate
// accumulators that are added last; fixed and position independent.
// ------------------------------------------------------------------------------------------------
typedef float e_f32x4v __attribute__((ext_vector_type(4)));
#define H13_TR 8
#define H13_TC 16
#define H13_S (H13_TC + 8)                  // staged row: columns gx0 - 4 .. gx0 + 19 as six aligned float4
#define H13_CS ((H13_TR + 2) * H13_S)      // 240 floats per channel = 16 mod 32: the four k rows of a B read sit 16 banks apart
#define H13_KC 16
#define H13_UNITS (H13_KC * (H13_TR + 2) * (H13_S / 4))     // float4 moves per chunk: 960

__global__ __launch_bounds__(256) void deconv5_cout3_mfma_kernel(const ConvArgs a, int tiles_x, int tiles_y, int tpw,
                                                                 unsigned long long* prof) {
#ifdef H13_PROF
    unsigned long long tq[8], tp[4];
    tq[0] = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NCH = 64 / H13_KC;
    constexpr unsigned OOB = 0x80000000u;          // buffer offset past any record: the load returns 0
    __shared__ float af[16 * 9 * 64];
    __shared__ __attribute__((aligned(16))) float tl[2][H13_KC * H13_CS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work-group = a run of up to tpw x-adjacent tiles of one tile row
    const int strips = (tiles_x + tpw - 1) / tpw;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int sx = b % strips; b /= strips;
    const int ty = b % tiles_y; const int n = b / tiles_y;
    const int tx0 = sx * tpw, nt = min(tpw, tiles_x - tx0);
    const int gy0 = ty * H13_TR;
    const int HW = a.H * a.W;
    const int j = lane & 15, kg = lane >> 4;
#ifdef H13_PROF
    __builtin_amdgcn_sched_barrier(0);
    tp[3] = __builtin_amdgcn_s_memtime() + (unsigned long long)(nt + HW == -12345);
    __builtin_amdgcn_sched_barrier(0);
#endif
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * 64 * HW), 0, 64 * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 25 * a.Cout * 64 * 4, 0x00020000);

    // input staging: a chunk is 16 channel slots x 10 rows x 6 float4 = 960 moves, <= 4 per thread, the same for every
    // chunk of a tile (scalar chunk offset).  Slot sl = 4cc + kg holds channel 16kg + 4ch + cc: lane group kg of the
    // MFMA walks the contiguous channels 16kg .. 16kg+15, so the filter gather below can use 16-byte loads.  Rows
    // outside the image read 0 through the OOB offset; a float4 that straddles the right edge is masked per element.
    constexpr int NU = (H13_UNITS + 255) / 256;
    int ubase[NU], ulds[NU], ucol[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + 256 * i;
        const int sl = u / 60, rem = u - sl * 60;
        const int row = rem / 6, q = rem - row * 6;
        const int iy = gy0 - 1 + row;
        const bool ok = u < H13_UNITS && iy >= 0 && iy < a.H;
        ubase[i] = ok ? (16 * (sl & 3) + (sl >> 2)) * HW + iy * a.W : -1;
        ucol[i] = 4 * q - 4;
        ulds[i] = u < H13_UNITS ? sl * H13_CS + row * H13_S + 4 * q : -1;
    }
    unsigned voff[NU], vmask = 0;
    auto tile_off = [&](int gx0) {
        vmask = 0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int ix = gx0 + ucol[i];
            voff[i] = ubase[i] >= 0 && ix >= 0 && ix < a.W ? (unsigned)(ubase[i] + ix) * 4u : OOB;
            const int left = a.W - ix;                                   // elements of this float4 inside the row
            vmask |= (left >= 4 ? 15u : (1u << max(left, 0)) - 1u) << (4 * i);
        }
    };
    auto request = [&](e_f32x4* dst, int chunk) {
        const int cb = chunk * 4 * HW * 4;
#pragma unroll
        for (int i = 0; i < NU; ++i) dst[i] = __builtin_bit_cast(e_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[i], cb, 0));
    };
    auto deposit = [&](float* dstl, const e_f32x4* src, unsigned m) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            e_f32x4 v = src[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (m >> (4 * i + e)) & 1 ? v[e] : 0.f;
            if (i + 1 < NU || ulds[i] >= 0) *reinterpret_cast<e_f32x4*>(dstl + ulds[i]) = v;
        }
    };

    // filter fragments af[(c*9 + t)*64 + lane]: lane (row i = (co, py, px), k row kg), k-step c, neighbour t = (dy, dx)
    // holds the tap ky = py + 1 - 2dy, kx = px + 1 - 2dx of channel 16kg + c; wave w gathers steps 4w..4w+3 as ONE
    // 16-byte load per neighbour.  The tap index is linear in the lane's (py, px): one lane base + uniform offsets; it
    // falls outside the filter only for dy = +1 on py = 0 (dx alike)
    // (gathered first: it is an L2 hit that would otherwise queue behind the tile's HBM reads)
    e_f32x4 fv[9];
    {
        const int i = lane & 15, co = i >> 2, py = (i >> 1) & 1, px = i & 1;
        const int tapf = a.Cout * 64;                                  // floats per filter tap
        const unsigned lbase = co < a.Cout ? (unsigned)((((py + 1) * 5 + px + 1) * a.Cout + co) * 64 + 16 * kg + 4 * wave) * 4u : OOB;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = !(dy == 1 && py == 0) && !(dx == 1 && px == 0);
            const int so = (-10 * dy - 2 * dx) * tapf * 4;
            // lbase + so >= 0 for every valid tap; an invalid one is redirected to the OOB offset
            fv[t] = __builtin_bit_cast(e_f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (dy == 1 || dx == 1) && !ok ? OOB : lbase + (unsigned)so, 0, 0));
        }
    }
    // the input is requested TWO chunks ahead (two register sets, two LDS buffers): one chunk of MFMAs (1.2 us) is
    // shorter than an HBM round trip under load
    tile_off(tx0 * H13_TC);
    unsigned vm_cur = vmask;
    e_f32x4 st[2][NU];
    request(st[0], 0);
    request(st[1], 1);
#ifdef H13_PROF
    tp[0] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) af[((4 * wave + c) * 9 + t) * 64 + lane] = fv[t][c];
#ifdef H13_PROF
    tp[1] = __builtin_amdgcn_s_memtime();
#endif
    deposit(tl[0], st[0], vm_cur);
#ifdef H13_PROF
    tp[2] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef H13_PROF
    tq[1] = __builtin_amdgcn_s_memtime();
#endif

    const int co = kg;
    float sc = 0.f, sh = 0.f, om = 0.f, os = 1.f;
    const bool dn = a.out_mean != nullptr || (a.builtin_norm & 2);
    if (co < a.Cout) {
        sc = a.scale[co]; sh = a.shift[co];
        if (a.out_mean) { om = a.out_mean[co]; os = a.out_std[co]; }
        else if (a.builtin_norm & 2) { om = IC_IMG_MEAN[co]; os = IC_IMG_STD[co]; }
    }
    const size_t ohw = (size_t)a.OH * a.OW;

    for (int tile = 0; tile < nt; ++tile) {
        const int gx0 = (tx0 + tile) * H13_TC;
        e_f32x4v acc[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][h][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            // request the chunk two stages ahead into the register set whose content went to LDS one stage ago
            if (ch + 2 < NCH) {
                request(st[ch & 1], ch + 2);
            } else if (tile + 1 < nt) {
                if (ch + 2 == NCH) tile_off(gx0 + H13_TC);
                request(st[ch & 1], ch + 2 - NCH);
            }
            const bool more = ch + 1 < NCH || tile + 1 < nt;
            const float* __restrict__ Lb = tl[ch & 1] + kg * H13_CS + (2 * wave + 1) * H13_S + j + 4;
            const float* __restrict__ ab = af + (ch * (H13_KC / 4) * 9) * 64 + lane;
            // operands of channel group cc + 1 are requested before the MFMAs of group cc (the compiler would
            // otherwise read each one right before its use and wait): A = 9 neighbours, B = 4 rows x 3 columns
            // shared by the wave's two grid rows
            float aq[2][9], bq[2][12];
            auto fetch = [&](int cc, float* av, float* bv) {
#pragma unroll
                for (int t = 0; t < 9; ++t) av[t] = ab[(cc * 9 + t) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int d = 0; d < 3; ++d) bv[r * 3 + d] = Lb[4 * cc * H13_CS + (r - 1) * H13_S + (d - 1)];
            };
            fetch(0, aq[0], bq[0]);
#pragma unroll
            for (int cc = 0; cc < H13_KC / 4; ++cc) {
                if (cc + 1 < H13_KC / 4) fetch(cc + 1, aq[(cc + 1) & 1], bq[(cc + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float av = aq[cc & 1][t];
                    acc[0][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[cc & 1][t], acc[0][t & 1], 0, 0, 0);
                    acc[1][t & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[cc & 1][t + 3], acc[1][t & 1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
                if (ch + 1 == NCH) vm_cur = vmask;                      // the stage being written opens the next tile
                deposit(tl[(ch & 1) ^ 1], st[(ch & 1) ^ 1], vm_cur);
            }
            __syncthreads();
#ifdef H13_PROF
            if (tile == 0) tq[2 + ch] = __builtin_amdgcn_s_memtime();
#endif
        }

        // lane (kg, j): output channel kg, grid position (gy, gx0 + j); acc register r = phase (py = r >> 1, px = r & 1)
        const int gx = gx0 + j;
        if (co < a.Cout && gx < a.W) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int gy = gy0 + 2 * wave + g;
                if (gy >= a.H) break;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(acc[g][0][r] + acc[g][1][r], sc, sh);
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (dn) v = fminf(fmaxf(__fadd_rn(__fmul_rn(v, os), om), 0.f), 255.f);
                    else if (a.builtin_norm & 4) v = fminf(fmaxf(v, 0.f), 255.f);
                    o[r] = v;
                }
                float* dst = a.y + ((size_t)n * a.Cout + co) * ohw + (size_t)(2 * gy) * a.OW + 2 * gx;
                *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
                *reinterpret_cast<float2*>(dst + a.OW) = make_float2(o[2], o[3]);
            }
        }
#ifdef H13_PROF
        if (tile == 0) tq[6] = __builtin_amdgcn_s_memtime();
#endif
    }
#ifdef H13_PROF
    if (prof && (tid & 63) == 0)
        for (int i = 0; i < 8; ++i) prof[((size_t)blockIdx.x * 4 + wave) * 8 + i] = i == 7 ? (tp[0] - tq[0]) | ((tp[1] - tp[0]) << 20) | ((tp[2] - tp[1]) << 40) : i == 6 ? tp[3] - tq[0] : tq[i];
#endif
}

// tiles per work-group: automatic unless the caller asks for a run length (ConvArgs::tune, from the per-call flags
// IC_EDGE_TILES_PER_WG(n) of ic_deconv2d_bn_act_f32 -- tests); the stamp buffer exists in -DH13_PROF builds only.
#ifdef H13_PROF
static unsigned long long* g_h13_prof = nullptr;
extern "C" void ic_edge_debug_set_prof_buffer(void* p) { g_h13_prof = (unsigned long long*)p; }
#else
static constexpr unsigned long long* g_h13_prof = nullptr;
#endif

int icx_deconv5_cout3_mfma(const ConvArgs& a, hipStream_t st) {
    if (a.KH != 5 || a.KW != 5 || a.Cout > 4 || a.Cin != 64 || a.pt != 1 || a.pl != 1 || a.res1 || a.res2 || a.in_mean ||
        (a.builtin_norm & 1) || a.w_sci != 1 || a.w_sco != 64)
        return IC_ERR_UNSUPPORTED;
    if ((long long)a.H * a.W * 64 * 4 >= (1ll << 31)) return IC_ERR_UNSUPPORTED;     // 32-bit buffer offsets
    const int tiles_x = ic_cdiv(a.W, H13_TC), tiles_y = ic_cdiv(a.H, H13_TR);
    // runs of x-adjacent tiles per work-group amortise the prologue (filter gather, first HBM round trip); the run
    // length is chosen by a small occupancy model: two work-groups fit a CU (LDS) and then share its matrix pipes
    int tpw = a.tune;
    if (tpw <= 0) {
        const double Pa = 6, Ta = 13, Pc = 12, Tc = 26;       // prologue / tile cost alone on a CU and co-resident (k clocks)
        double best = 1e30;
        for (int cand = 1; cand <= std::min(tiles_x, 16); ++cand) {
            const long long wgs = (long long)ic_cdiv(tiles_x, cand) * tiles_y * a.N;
            const int run = ic_cdiv(tiles_x, ic_cdiv(tiles_x, cand));
            const long long full = wgs / 512, rem = wgs % 512;
            double cost = full * (Pc + run * Tc);
            if (rem > 0) cost += rem <= 256 && full == 0 ? Pa + run * Ta : rem <= 256 ? 0.5 * (Pc + run * Tc) : Pc + run * Tc;
            if (cost < best) { best = cost; tpw = cand; }
        }
    }
    const int strips = ic_cdiv(tiles_x, tpw);
    tpw = ic_cdiv(tiles_x, strips);                  // even out the runs
    hipLaunchKernelGGL(deconv5_cout3_mfma_kernel, dim3((unsigned)(strips * tiles_y * a.N)), dim3(256), 0, st, a, tiles_x,
                       tiles_y, tpw, g_h13_prof);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
