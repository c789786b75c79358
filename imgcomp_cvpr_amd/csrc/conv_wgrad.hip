// Filter gradients of every convolution / transposed convolution of the autoencoder on the fp32 matrix cores.
//   reference: the backward of slim.conv2d / slim.conv2d_transpose at code/autoencoder.py:222-265,285 that
//   tf.gradients builds for train.py:339-349.
// One generic form covers all 70 layers:
//     dW[t][a][b] = sum_{n,qy,qx} U[n][a][s*qy + ty + oy0][s*qx + tx + ox0] * V[n][b][qy][qx]      t = (ty,tx)
//   * conv (stride s, TF SAME pads):      U = layer input x (A = Cin),        V = dy (B = Cout)  -> dW[kh][kw][cin][cout]
//   * transposed conv (stride 2):         U = dy on the big grid (A = Cout),  V = x (B = Cin)    -> dW[kh][kw][cout][cin]
//   i.e. exactly the TF variable layouts, so the optimiser updates the checkpoint-layout tensors in place.
// Per tap this is a GEMM with K = the N*VH*VW positions: D[a][b] += A_op[a][k] * B_op[k][b] on
// v_mfma_f32_32x32x2_f32.  Both operands are channel-major in HBM (NCHW), i.e. k runs along the contiguous
// axis, so a 32-position chunk of U (shifted by the tap) and V is staged through LDS as [channel][33] rows
// (coalesced global reads along positions, conflict-free LDS reads along channels).
// K is split over the grid (S slices of the position range); every work-group writes its partial [A][B] tile set
// and ic_conv2d_wgrad_f32 reduces the slices in fixed order (deterministic; no atomics).
#include "internal.h"

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef WG_KP
#define WG_KP 32            // positions per LDS chunk (16 was tried: 4 resident groups per CU but twice the slices to reduce -- slower)
#endif
#define WG_LS (WG_KP + 4)    // LDS row stride (floats): 16-byte aligned rows; 20 or 36 * row mod 64 spreads 16 rows over all banks

struct WgArgs {
    const float* U; const float* V; float* partial;
    int N, A, UH, UW, B, VH, VW;
    int KH, KW, stride, oy0, ox0;
    int P, PS, S;           // positions, positions per slice, slices
    // 3-D mode (context model, VALID masked conv3d): U (N,A,VD+1,VH+2,VW+2), V (N,B,VD,VH,VW); tap = index into taps[]
    int VD, UD;
    int taps[18];           // live tap -> kd*9 + kh*3 + kw
    const float* q; int qC, qh, qw; float pad_value;   // A == 1 and q != null: U is the padded symbol volume (pad-on-load)
};

template <int TA, int TB, int WA, int WB, bool MODE3D>
__global__ __launch_bounds__(64 * WA * WB) void conv_wgrad_kernel(const WgArgs a) {
    constexpr int NTH = 64 * WA * WB;
    constexpr int AT = 32 * TA * WA, BT = 32 * TB * WB;       // channel rows staged per chunk
    constexpr int ROWS = AT + BT;
    constexpr int RPT = ROWS / (NTH / WG_KP);                  // rows each thread stages (its pixel column is fixed)
    static_assert(ROWS % (NTH / WG_KP) == 0, "rows divide evenly over the thread rows");
    __shared__ __attribute__((aligned(16))) float lds[2][ROWS * WG_LS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave % WA, wb = wave / WA;
    const int split = blockIdx.x, tap = blockIdx.y;
    const int nbg = (a.B + BT - 1) / BT;
    const int a0 = (blockIdx.z / nbg) * AT, b0 = (blockIdx.z % nbg) * BT;
    const int ty = MODE3D ? 0 : tap / a.KW, tx = MODE3D ? 0 : tap % a.KW;
    const int q_begin = split * a.PS, q_end = min(q_begin + a.PS, a.P);
    const int VHW = a.VH * a.VW, UHW = a.UH * a.UW;
    const int px = tid % WG_KP, trow = tid / WG_KP;            // this thread's pixel column and first row

    wg_f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int TROWS = NTH / WG_KP;                         // thread rows; AT and BT are multiples of 32 >= TROWS
    constexpr int RA = AT / TROWS, RB = BT / TROWS;            // U rows and V rows staged by each thread
    static_assert(RA + RB == RPT, "row split");
    float st[RPT];
    // Staging loads are raw buffer loads: per-lane byte offset of the position (row 0 of this thread) + a SCALAR offset
    // per staged row (k * TROWS channels further) -- no vector address arithmetic per row; padded positions, dead lanes and
    // channels past A / B get an out-of-range lane offset and the hardware returns 0.  (The first version spent 3.7 vector
    // instructions per MFMA on addresses and selects; vector instructions cost matrix-pipe issue time.)
    const int uvol = MODE3D ? a.UD * UHW : UHW, vvol = MODE3D ? a.VD * VHW : VHW;
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, MODE3D && a.q ? 0 : (int)((long long)a.N * a.A * uvol * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc((void*)a.V, 0, (int)((long long)a.N * a.B * vvol * 4), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const bool a_full = a0 + AT <= a.A, b_full = b0 + BT <= a.B;           // work-group uniform
    auto fetch = [&](int q0) {
        const int q = q0 + px;
        const bool live = q < q_end;
        const int qc = live ? q : q_begin;
        unsigned uoff, voff;
        float qv = 0.f;
        if (!MODE3D) {
            const int n = qc / VHW, rem = qc - n * VHW;
            const int qy = rem / a.VW, qx = rem - qy * a.VW;
            const int iy = a.stride * qy + ty + a.oy0, ix = a.stride * qx + tx + a.ox0;
            const bool uok = live && iy >= 0 && iy < a.UH && ix >= 0 && ix < a.UW;
            uoff = uok ? (unsigned)(((n * a.A + a0 + trow) * UHW + iy * a.UW + ix) * 4) : OOB;
            voff = live ? (unsigned)(((n * a.B + b0 + trow) * VHW + rem) * 4) : OOB;
        } else {
            // position = (n, d, y, x) of the conv3d OUTPUT volume; the tap offset is never out of range (VALID)
            const int n = qc / vvol, rem = qc - n * vvol;
            const int d = rem / VHW, r2 = rem - d * VHW;
            const int y = r2 / a.VW, x = r2 - y * a.VW;
            const int t3 = a.taps[tap];
            const int ud = d + t3 / 9, uy = y + (t3 % 9) / 3, ux = x + t3 % 3;
            uoff = live ? (unsigned)(((n * a.A + a0 + trow) * uvol + ud * UHW + uy * a.UW + ux) * 4) : OOB;
            voff = live ? (unsigned)(((n * a.B + b0 + trow) * vvol + rem) * 4) : OOB;
            if (a.q) {                                 // U = symbol volume padded on load (depth front 4, H/W 4 each side)
                const int c = ud - 4, yy = uy - 4, xx = ux - 4;
                const bool in = c >= 0 && yy >= 0 && yy < a.qh && xx >= 0 && xx < a.qw;
                qv = !live ? 0.f : (in ? a.q[(((size_t)n * a.qC + c) * a.qh + yy) * a.qw + xx] : a.pad_value);
            }
        }
#pragma unroll
        for (int k = 0; k < RA; ++k) {
            if (MODE3D && a.q) { st[k] = (a0 + trow + k * TROWS < a.A) ? qv : 0.f; continue; }
            const unsigned vo = (a_full || a0 + trow + k * TROWS < a.A) ? uoff : OOB;
            st[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ur, vo, k * TROWS * uvol * 4, 0));
        }
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const unsigned vo = (b_full || b0 + trow + k * TROWS < a.B) ? voff : OOB;
            st[RA + k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vr, vo, k * TROWS * vvol * 4, 0));
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) lds[buf][(trow + k * TROWS) * WG_LS + px] = st[k];
    };

    fetch(q_begin);
    stash(0);
    __syncthreads();
    const int kh = lane >> 5, li = lane & 31;
    int buf = 0;
    for (int q0 = q_begin; q0 < q_end; q0 += WG_KP) {
        const bool more = q0 + WG_KP < q_end;
        if (more) fetch(q0 + WG_KP);
        const float* __restrict__ L = lds[buf];
        // k-step order inside the chunk: k-step 4 j4 + e multiplies positions (4 j4 + e, 16 + 4 j4 + e) -- the MFMA's two
        // k lanes take position p and p + 16 instead of 2 ks and 2 ks + 1 (any pairing sums the same products), so a
        // lane's operands of four consecutive k-steps are 16 contiguous bytes: one ds_read_b128 instead of four b32.
#pragma unroll
        for (int j4 = 0; j4 < WG_KP / 8; ++j4) {
            f32x4 av[TA], bv[TB];
#pragma unroll
            for (int i = 0; i < TA; ++i)
                av[i] = *(const f32x4*)(L + (32 * (TA * wa + i) + li) * WG_LS + (WG_KP / 2) * kh + 4 * j4);
#pragma unroll
            for (int j = 0; j < TB; ++j)
                bv[j] = *(const f32x4*)(L + (AT + 32 * (TB * wb + j) + li) * WG_LS + (WG_KP / 2) * kh + 4 * j4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < TB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
    }
    // partial[split][tap][a][b]; D[i][j]: i = (r&3) + 8*(r>>2) + 4*kh is the U channel, j = lane&31 the V channel
    float* __restrict__ out = a.partial + ((size_t)split * gridDim.y + tap) * a.A * a.B;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int cb = b0 + 32 * (TB * wb + j) + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ca = a0 + 32 * (TA * wa + i) + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (ca < a.A && cb < a.B) out[(size_t)ca * a.B + cb] = acc[i][j][r];
            }
        }
}

// dW[e] = sum_s partial[s][e] (+ wd * w[e]); slices summed in index order
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int S, long long count,
                                                           const float* __restrict__ w, float wd, float* __restrict__ dw) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double s = 0.0;                                              // the slices' fp32 sums are added up in float64 (fixed order)
    for (int k = 0; k < S; ++k) s += (double)partial[(size_t)k * count + e];
    if (w) s += (double)wd * (double)w[e];
    dw[e] = (float)s;
}

static void wg_plan(int A, int B, long long P, int KH, int KW, int* TA, int* WA, int* TB, int* WB, int* S, int* PS) {
    const int ta_tiles = ic_cdiv(A, 32), tb_tiles = ic_cdiv(B, 32);
    *TA = ta_tiles >= 2 ? 2 : 1; *WA = ta_tiles > 2 ? 2 : 1;
    *TB = tb_tiles >= 2 ? 2 : 1; *WB = tb_tiles > 2 ? 2 : 1;
    const int groups = ic_cdiv(A, 32 * *TA * *WA) * ic_cdiv(B, 32 * *TB * *WB);
    // one full round of resident work-groups: a second, partly filled round costs a whole work-group time
    // (711 work-groups on 512 slots ran as long as 1024 would have).  Residency is bounded by the LDS double buffer
    // and by ~4 waves per SIMD of registers.
    const int waves = *WA * *WB;
    const int lds_bytes = 2 * 32 * (*TA * *WA + *TB * *WB) * WG_LS * 4;
    int per_cu = 160 * 1024 / lds_bytes;
    if (per_cu > 16 / waves) per_cu = 16 / waves;
    if (per_cu < 1) per_cu = 1;
    long long want = (256ll * per_cu) / ((long long)KH * KW * groups);
    long long maxs = (P + 4 * WG_KP - 1) / (4 * WG_KP);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    long long ps = ((P + want - 1) / want + WG_KP - 1) / WG_KP * WG_KP;
    *PS = (int)ps;
    *S = (int)((P + ps - 1) / ps);
}

extern "C" size_t ic_conv2d_wgrad_workspace_bytes(int N, int A, int B, int VH, int VW, int KH, int KW) {
    if (N <= 0 || A <= 0 || B <= 0 || VH <= 0 || VW <= 0 || KH <= 0 || KW <= 0) return 0;
    int TA, WA, TB, WB, S, PS;
    wg_plan(A, B, (long long)N * VH * VW, KH, KW, &TA, &WA, &TB, &WB, &S, &PS);
    return (size_t)S * KH * KW * A * B * sizeof(float);
}

#define WG_LAUNCH(TA_, TB_, WA_, WB_)                                                                       \
    hipLaunchKernelGGL((conv_wgrad_kernel<TA_, TB_, WA_, WB_, false>),                                      \
                       dim3(S, KH * KW, ic_cdiv(A, 32 * TA_ * WA_) * ic_cdiv(B, 32 * TB_ * WB_)),           \
                       dim3(64 * WA_ * WB_), 0, st, a)

// U: (N,A,UH,UW) large-grid tensor, V: (N,B,VH,VW) small-grid tensor (VH = ceil(UH/stride)); see the header comment.
// w/wd: optional weight-decay term added to the result (dW += wd * w: slim.l2_regularizer, autoencoder.py:101-102).
extern "C" int ic_conv2d_wgrad_f32(const float* U, const float* V, float* dw, int N, int A, int UH, int UW, int B,
                                   int KH, int KW, int stride, const float* w, float wd,
                                   void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(U && V && dw && workspace && N > 0 && A > 0 && B > 0 && UH > 0 && UW > 0 && KH > 0 && KW > 0);
    if (stride != 1 && stride != 2) return IC_ERR_UNSUPPORTED;
    const int VH = ic_cdiv(UH, stride), VW = ic_cdiv(UW, stride);
    const long long P = (long long)N * VH * VW;
    // byte offsets into U and V are 31-bit (buffer loads; 2^31 is the out-of-range marker)
    if (P >= (1ll << 31) || (long long)N * A * UH * UW * 4 >= (1ll << 31) || (long long)N * B * VH * VW * 4 >= (1ll << 31))
        return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_conv2d_wgrad_workspace_bytes(N, A, B, VH, VW, KH, KW)) return IC_ERR_WORKSPACE;
    int TA, WA, TB, WB, S, PS;
    wg_plan(A, B, P, KH, KW, &TA, &WA, &TB, &WB, &S, &PS);
    WgArgs a{};
    a.U = U; a.V = V; a.partial = (float*)workspace;
    a.N = N; a.A = A; a.UH = UH; a.UW = UW; a.B = B; a.VH = VH; a.VW = VW;
    a.KH = KH; a.KW = KW; a.stride = stride;
    a.oy0 = -ic_same_pad_before(UH, KH, stride); a.ox0 = -ic_same_pad_before(UW, KW, stride);
    a.P = (int)P; a.PS = PS; a.S = S;
    hipStream_t st = (hipStream_t)stream;
    const int key = TA * 1000 + TB * 100 + WA * 10 + WB;
    switch (key) {
        case 2222: WG_LAUNCH(2, 2, 2, 2); break;
        case 2212: WG_LAUNCH(2, 2, 1, 2); break;
        case 2221: WG_LAUNCH(2, 2, 2, 1); break;
        case 2211: WG_LAUNCH(2, 2, 1, 1); break;
        case 2121: WG_LAUNCH(2, 1, 2, 1); break;
        case 2111: WG_LAUNCH(2, 1, 1, 1); break;
        case 1212: WG_LAUNCH(1, 2, 1, 2); break;
        case 1211: WG_LAUNCH(1, 2, 1, 1); break;
        case 1111: WG_LAUNCH(1, 1, 1, 1); break;
        default: return IC_ERR_UNSUPPORTED;
    }
    const long long count = (long long)KH * KW * A * B;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, a.partial, S, count,
                       w, wd, dw);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// dW[live tap][a][b] of a masked VALID (2,3,3) conv3d (probclass.py:227-261): scattered into the TF layout
// [2][3][3][A][B] with the dead taps left at zero.
__global__ __launch_bounds__(256) void wgrad3d_reduce_kernel(const float* __restrict__ partial, int S, int NT, int AB,
                                                             WgArgs a, float* __restrict__ dw) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)18 * AB) return;
    const int t3 = (int)(e / AB), r = (int)(e % AB);
    int lt = -1;
    for (int k = 0; k < NT; ++k) if (a.taps[k] == t3) lt = k;
    double s = 0.0;
    if (lt >= 0) for (int k = 0; k < S; ++k) s += (double)partial[((size_t)k * NT + lt) * AB + r];
    dw[e] = (float)s;
}

extern "C" size_t ic_pc_wgrad_workspace_bytes(int N, int A, int B, int VD, int VH, int VW) {
    if (N <= 0 || A <= 0 || B <= 0 || VD <= 0 || VH <= 0 || VW <= 0) return 0;
    int TA, WA, TB, WB, S, PS;
    wg_plan(A, B, (long long)N * VD * VH * VW, 2, 7, &TA, &WA, &TB, &WB, &S, &PS);
    return (size_t)S * 14 * A * B * sizeof(float);
}

#define WG3_LAUNCH(TA_, TB_, WA_, WB_)                                                                      \
    hipLaunchKernelGGL((conv_wgrad_kernel<TA_, TB_, WA_, WB_, true>),                                       \
                       dim3(S, NT, ic_cdiv(A, 32 * TA_ * WA_) * ic_cdiv(B, 32 * TB_ * WB_)),                \
                       dim3(64 * WA_ * WB_), 0, st, a)

// U: (N,A,VD+1,VH+2,VW+2) layer input, or -- when q != NULL and A == 1 -- the symbol volume q (N,VD-3... see header)
extern "C" int ic_pc_wgrad_f32(const float* U, const float* q, float pad_value, const float* V, float* dw,
                               int N, int A, int B, int VD, int VH, int VW, int first_mask,
                               void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG((U || q) && V && dw && workspace && N > 0 && A > 0 && B > 0 && VD > 0 && VH > 0 && VW > 0);
    if (q && A != 1) return IC_ERR_ARG;
    const long long P = (long long)N * VD * VH * VW;
    if (P >= (1ll << 31) || P * B * 4 >= (1ll << 31) ||
        (long long)N * A * (VD + 1) * (VH + 2) * (VW + 2) * 4 >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_pc_wgrad_workspace_bytes(N, A, B, VD, VH, VW)) return IC_ERR_WORKSPACE;
    int TA, WA, TB, WB, S, PS;
    wg_plan(A, B, P, 2, 7, &TA, &WA, &TB, &WB, &S, &PS);
    WgArgs a{};
    a.U = U; a.V = V; a.partial = (float*)workspace;
    a.N = N; a.A = A; a.B = B; a.VD = VD; a.VH = VH; a.VW = VW; a.UD = VD + 1; a.UH = VH + 2; a.UW = VW + 2;
    a.KH = 1; a.KW = 1; a.stride = 1; a.P = (int)P; a.PS = PS; a.S = S;
    a.q = q; a.pad_value = pad_value;
    if (q) { a.qC = VD + 1 - 4; a.qh = VH + 2 - 8; a.qw = VW + 2 - 8; }
    int NT = 0;
    for (int t3 = 0; t3 < 18; ++t3) {
        const int kd = t3 / 9, kh = (t3 % 9) / 3, kw = t3 % 3;
        const bool dead = kd == 1 && (kh == 2 || (kh == 1 && (first_mask ? kw >= 1 : kw >= 2)));
        if (!dead) a.taps[NT++] = t3;
    }
    hipStream_t st = (hipStream_t)stream;
    const int key = TA * 1000 + TB * 100 + WA * 10 + WB;
    switch (key) {
        case 2222: WG3_LAUNCH(2, 2, 2, 2); break;
        case 2212: WG3_LAUNCH(2, 2, 1, 2); break;
        case 2221: WG3_LAUNCH(2, 2, 2, 1); break;
        case 2211: WG3_LAUNCH(2, 2, 1, 1); break;
        case 2121: WG3_LAUNCH(2, 1, 2, 1); break;
        case 2111: WG3_LAUNCH(2, 1, 1, 1); break;
        case 1212: WG3_LAUNCH(1, 2, 1, 2); break;
        case 1211: WG3_LAUNCH(1, 2, 1, 1); break;
        case 1111: WG3_LAUNCH(1, 1, 1, 1); break;
        default: return IC_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(wgrad3d_reduce_kernel, dim3(ic_cdiv(18 * A * B, 256)), dim3(256), 0, st, a.partial, S, NT, A * B, a, dw);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
