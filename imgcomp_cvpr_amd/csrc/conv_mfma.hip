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
    const float* x; const float* scale; const float* shift; float* y;
    int N, IH, IW;          // input extent
    int GH, GW;             // output grid extent (per phase for transposed)
    int OH, OW, OS;         // output tensor extent, grid -> output stride (2 for transposed phases)
    int Cout, relu;
    int tiles_x, tiles_y;
    int ncot;               // 32-channel tiles in the packed filter (its stride); the launch may cover fewer (live tiles only)
    unsigned long long* prof;   // tuning builds (CM_PROF) only
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

struct GPhase { const float* wp; int oy0, ox0, py, px; };

// CPC: 8-channel groups staged per chunk (one barrier per chunk: more groups = more MFMAs between barriers)
template <int NTY, int NTX, int PS, int WK, int TR, int TC, int CPC = 1>
struct GGeo {
    static constexpr int ROWS = (TR - 1) * PS + NTY, COLS = (TC - 1) * PS + NTX;
    static constexpr int S = COLS, CS = ROWS * COLS, CHUNK = GKC * WK * CS * CPC;
    static constexpr int NST = (CHUNK + 255) / 256;
    static constexpr int LDSF = 2 * NST * 256;
    static_assert(NST <= 32, "in-bounds mask is 32 bits");
};

// WM x WN x WK = 4 waves: WM output-channel tiles x WN pixel groups x WK slices of the K (input channel)
// axis.  With WK > 1 each stage holds 8*WK channels in LDS, wave wk multiplies channels [8 wk, 8 wk + 8) of
// every stage, and the WK partial accumulators are summed through LDS in the fixed order wk = 0..WK-1
// (deterministic, position independent).  Split-K is for layers whose output is too small to fill the chip
// with whole-K tiles (to_bn: 6144 pixels x 33 channels, K = 3200).
// KEEP: instead of storing, leave the finished values (BN scale / shift and activation applied) in keep[PT]: the caller pairs
// the two x-phases of a transposed convolution and writes them as whole 8-byte runs (deconv5_pair_kernel).
// X1: Cout = 32 m + 1 (to_bn with the importance map: C + 1 = 33 channels).  The one channel beyond the last full tile does
// not get a 32-row matrix tile of its own (31 of 32 rows zero, a second work-group per pixel tile: half of the layer's matrix
// time in round 2): every wave also accumulates it on the vector unit, one fma per k-step next to the tile's MFMA -- the B
// operand is already in the register, the weight comes as one 16-byte load per tap from that channel's row of the packed
// fragments.  Sum order of that channel: per (K-slice, k parity) ascending chains, then parity 0 + parity 1, then the K-slices
// in order -- fixed and position independent like the tiles' own.
template <int NTY, int NTX, int PS, int CIN, int WM, int WN, int WK, int PT, int TR, int TC, int RD, bool KEEP = false, bool X1 = false, int CPC = 1>
__device__ __forceinline__ void conv_mfma_body(const GArgs& a, const GPhase& ph, float* __restrict__ lds, f32x16* keep = nullptr) {
#ifdef CM_PROF
    const unsigned long long cm_t0 = __builtin_amdgcn_s_memtime();
#endif
    using G = GGeo<NTY, NTX, PS, WK, TR, TC, CPC>;
    constexpr int NT = NTY * NTX;
    constexpr int NCH = CIN / (GKC * WK * CPC);
    static_assert(CPC == 1 || WK == 1, "several channel groups per chunk: whole-K waves only");
    static_assert(CIN % (GKC * WK * CPC) == 0, "Cin multiple of the chunk");
    constexpr int S = G::S, CS = G::CS, CHUNK = G::CHUNK, NST = G::NST;
    static_assert(WM * WN * WK == 4, "4 waves per work-group");
    static_assert(TR * TC == 32 * PT * WN, "tile = PT accumulator tiles per wave x WN pixel groups");
    static_assert(NT % RD == 0 && RD >= 3, "ring depth must divide the tap count");
    static_assert(CIN % (GKC * WK) == 0, "Cin multiple of 8 * WK");
    static_assert(WK == 1 || (16 % WK == 0 && WM * WN * WK * PT * 16 * 64 <= G::LDSF), "reduction scratch fits in the staging buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);
    int b = ic_xcd_run(blockIdx.x, gridDim.x);             // contiguous runs of tiles per XCD: halo rows in one L2
    const int tx_ = b % a.tiles_x; b /= a.tiles_x;
    const int ty_ = b % a.tiles_y; const int n = b / a.tiles_y;
    const int cot = blockIdx.y * WM + wm;                 // this wave's 32-channel output tile
    const int ncot = a.ncot;
    const int gx0 = tx_ * TC, gy0 = ty_ * TR;
    const int IHW = a.IH * a.IW;
    const float* __restrict__ xin = a.x + (size_t)n * CIN * IHW;

    // staging plan
    int goff[NST];
    unsigned inb = 0;
    const int iy0 = gy0 * PS + ph.oy0, ix0 = gx0 * PS + ph.ox0;
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
        boff[p] = (GKC * wk + kh) * CS + (q / TC) * PS * S + (q % TC) * PS;
    }
    // A fragments of (8-channel chunk c8, tap t) for this wave's output channels sit at float4 index
    // ((c8*NT + t)*ncot + cot)*64 + lane: a buffer load with a per-lane offset that never changes and a SCALAR offset per
    // (chunk, tap) -- no vector address arithmetic in the loop.
    const int cot_u = __builtin_amdgcn_readfirstlane(cot), wk_u = __builtin_amdgcn_readfirstlane(wk);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)ph.wp, 0, (CIN / GKC) * NT * ncot * 256 * 4, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    const int wstep = ncot * 1024;                         // bytes per tap
    const int cstep = NT * wstep;                          // bytes per 8-channel chunk
    auto wload = [&](int c8, int t) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, c8 * cstep + t * wstep + cot_u * 1024, 0));
    };

    f32x16 acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // BN scale / shift of the channels this lane finishes, requested before the main loop: fetched inside the store loop,
    // each of the 16 channel iterations waited out its own L2 round trip (25 k of a wave's 165 k clocks in h2,
    // a -DCM_PROF build).  Channels past Cout read 0 through the descriptor's range check.
    constexpr int RPW = 16 / WK;                           // accumulator registers finished by each K-slice wave
    float bsc[RPW], bsh[RPW];
    {
        const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc((void*)a.scale, 0, a.Cout * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc((void*)a.shift, 0, a.Cout * 4, 0x00020000);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = (WK > 1 ? wk * RPW : 0) + rr;
            const unsigned co4 = (unsigned)(32 * cot + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 4u;
            bsc[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scr, co4, 0, 0));
            bsh[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shr, co4, 0, 0));
        }
    }
    f32x4 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wload(wk_u, t);
    // X1: the extra channel's weights for this lane's k parity -- row 0 of the fragments of tile (Cout - 1) / 32: floats
    // [0..3] (parity 0) and [128..131] (parity 1) of each 256-float fragment
    static_assert(!X1 || (PT == 1 && WM == 1 && WN == 1), "extra channel: one pixel tile per wave");
    const int cotx_u = X1 ? (a.Cout - 1) / 32 : 0;
    const unsigned xlane = (unsigned)(lane >> 5) * 512u;
    auto xload = [&](int c8, int t) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, xlane, c8 * cstep + t * wstep + cotx_u * 1024, 0));
    };
    f32x4 xring[X1 ? RD : 1];
    float accx = 0.f;
    if constexpr (X1) {
#pragma unroll
        for (int t = 0; t < RD - 2; ++t) xring[t] = xload(wk_u, t);
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const float v = xin[goff[i]];
        lds[tid + 256 * i] = ((inb >> i) & 1) ? v : 0.f;
    }
    __syncthreads();
#ifdef CM_PROF
    const unsigned long long cm_t1 = __builtin_amdgcn_s_memtime();
#endif

    for (int c = 0; c < NCH; ++c) {
        const bool more = c + 1 < NCH;
        float st[NST];
        if (more) {
            const float* xc = xin + (size_t)(c + 1) * GKC * WK * CPC * IHW;
#pragma unroll
            for (int i = 0; i < NST; ++i) st[i] = xc[goff[i]];
        }
        const float* __restrict__ L = lds + (c & 1) * (NST * 256);
        // the chunk's 8-channel groups one after the other as ONE sequence of CPC * NT taps (u = group * NT + tap): filter ring
        // and B-operand double buffer run across the group boundaries
        const int c8 = c * WK * CPC + wk_u, c8n = more ? (c + 1) * WK * CPC + wk_u : c8;   // past the end: re-read (never used)
        constexpr int NU = CPC * NT;
        // B operands of tap u + 1 are read while the MFMAs of tap u run (left alone the compiler reads each one right
        // before its use: load, wait, multiply); the first tap of a chunk reads its own after the barrier
        float bq[2][4][PT];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int p = 0; p < PT; ++p) bq[0][ks][p] = L[boff[p] + 2 * ks * CS];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            // ring: request tap u + RD - 2 into the slot whose last reader was tap u - 2
            {
                const int un = u + RD - 2;
                if (un < NU) ring[un % RD] = wload(c8 + un / NT, un % NT);
                else ring[un % RD] = wload(c8n + (un - NU) / NT, (un - NU) % NT);
                if constexpr (X1) {
                    if (un < NU) xring[un % RD] = xload(c8 + un / NT, un % NT);
                    else xring[un % RD] = xload(c8n + (un - NU) / NT, (un - NU) % NT);
                }
            }
            if (u + 1 < NU) {
                const int tn = (u + 1) % NT, gn = (u + 1) / NT;
                const int tapoff = (tn / NTX) * S + (tn % NTX) + 8 * gn * CS;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int p = 0; p < PT; ++p) bq[(u + 1) & 1][ks][p] = L[boff[p] + 2 * ks * CS + tapoff];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float av = ring[u % RD][ks];
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[u & 1][ks][p], acc[p], 0, 0, 0);
                // volatile asm: left to the compiler, the whole fma chain sinks behind the K loop and drags every B operand
                // of the chunk along in registers (256 VGPR + 112 AGPR, one wave per SIMD)
                if constexpr (X1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accx) : "v"(xring[u % RD][ks]), "v"(bq[u & 1][ks][0]));
            }
#pragma unroll
            for (int i = 0; i < 4 * PT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // 1 MFMA
                if (i == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // the tap's filter request
                if (X1 && i == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... and the extra channel's
                if (X1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);         // the extra channel's fma of this k-step
                if (u + 1 < NU) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // 1 LDS read of the next tap
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            float* __restrict__ Ln = lds + ((c & 1) ^ 1) * (NST * 256);
#pragma unroll
            for (int i = 0; i < NST; ++i) Ln[tid + 256 * i] = ((inb >> i) & 1) ? st[i] : 0.f;
        }
        // raw barrier: own LDS traffic done, filter prefetch stays in flight (no vmcnt(0) drain)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

#ifdef CM_PROF
    const unsigned long long cm_t2 = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*kh channel, j pixel
    float* __restrict__ y = a.y;
    const size_t OHW = (size_t)a.OH * a.OW;
    const size_t cbase = ((size_t)n * a.Cout + 32 * cot + 4 * kh) * OHW;
    if (WK > 1) {
        // all waves passed the loop's last barrier: the staging buffers are free
        float* red = lds + (size_t)((wm * WN + wn) * WK) * (PT * 16 * 64);
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(size_t)((wk * PT + p) * 16 + r) * 64 + lane] = acc[p][r];
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = wk * RPW + rr;
                float sum = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < WK; ++k2) sum += red[(size_t)((k2 * PT + p) * 16 + r) * 64 + lane];
                acc[p][rr] = sum;                          // compacted: slot rr now holds register wk*RPW + rr
            }
    }
    if constexpr (X1) {
        // pixel j of the tile: parity 0 + parity 1 (lanes j and j + 32), then the K-slices in order through LDS
        const float half_sum = accx + __shfl_xor(accx, 32);
        if (WK > 1) {
            __syncthreads();                               // the reduction scratch above is read: reuse its first floats
            if (lane < 32) lds[wk * 32 + lane] = half_sum;
            __syncthreads();
        }
        if (wk == 0 && lane < 32) {
            float sum = WK > 1 ? 0.f : half_sum;
            if (WK > 1) {
#pragma unroll
                for (int k2 = 0; k2 < WK; ++k2) sum += lds[k2 * 32 + lane];
            }
            const int cx = a.Cout - 1;
            const int q = lane, gy = gy0 + q / TC, gx = gx0 + q % TC;
            if (gy < a.GH && gx < a.GW) {
                float v = fmaf(sum, a.scale[cx], a.shift[cx]);
                if (a.relu) v = fmaxf(v, 0.f);
                y[((size_t)n * a.Cout + cx) * OHW + (size_t)(gy * a.OS + ph.py) * a.OW + (gx * a.OS + ph.px)] = v;
            }
        }
    }
    if constexpr (KEEP) {
        static_assert(!KEEP || WK == 1, "paired phases: whole-K waves");
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaf(acc[p][r], bsc[r], bsh[r]);
                keep[p][r] = a.relu ? fmaxf(v, 0.f) : v;
            }
        return;
    }
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * (PT * wn + p) + j;
        const int gy = gy0 + q / TC, gx = gx0 + q % TC;
        if (gy < a.GH && gx < a.GW) {
            const size_t pix = cbase + (size_t)(gy * a.OS + ph.py) * a.OW + (gx * a.OS + ph.px);
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = (WK > 1 ? wk * RPW : 0) + rr;
                const int crow = (r & 3) + 8 * (r >> 2);
                const int co = 32 * cot + crow + 4 * kh;
                if (co < (X1 ? a.Cout - 1 : a.Cout)) {
                    float v = fmaf(acc[p][rr], bsc[rr], bsh[rr]);
                    if (a.relu) v = fmaxf(v, 0.f);
                    y[pix + (size_t)crow * OHW] = v;
                }
            }
        }
    }
#ifdef CM_PROF
    if (a.prof && lane == 0) {
        unsigned long long* d = a.prof + 4 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave);
        d[0] = cm_t0; d[1] = cm_t1; d[2] = cm_t2; d[3] = __builtin_amdgcn_s_memtime();
    }
#endif
}

template <int NTY, int NTX, int PS, int CIN, int WM, int WN, int WK, int PT, int TR, int TC, int RD, int CPC = 1>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const GArgs a, const GPhase ph) {
    __shared__ float lds[GGeo<NTY, NTX, PS, WK, TR, TC, CPC>::LDSF];
    conv_mfma_body<NTY, NTX, PS, CIN, WM, WN, WK, PT, TR, TC, RD, false, false, CPC>(a, ph, lds);
}

template <int NTY, int NTX, int PS, int CIN, int WM, int WN, int WK, int PT, int TR, int TC, int RD>
__global__ __launch_bounds__(256) void conv_mfma_x1_kernel(const GArgs a, const GPhase ph) {
    __shared__ float lds[GGeo<NTY, NTX, PS, WK, TR, TC>::LDSF];
    conv_mfma_body<NTY, NTX, PS, CIN, WM, WN, WK, PT, TR, TC, RD, false, true>(a, ph, lds);
}

// all four output phases of a 5x5 / stride-2 transposed convolution in ONE launch (blockIdx.z = phase): 1.5 k
// work-groups of unequal length that the dispatcher packs onto the CUs as slots free up.
struct GPhases4 { GPhase p[4]; };
template <int CIN, int WM, int WN, int PT, int TR, int TC>
__global__ __launch_bounds__(256) void deconv5_mfma_kernel(const GArgs a, const GPhases4 ph) {
    __shared__ float lds[GGeo<3, 3, 1, 1, TR, TC>::LDSF];
    switch (blockIdx.z) {        // phase order (py,px) = (0,0),(0,1),(1,0),(1,1): taps 2x2, 2x3, 3x2, 3x3
        case 0: conv_mfma_body<2, 2, 1, CIN, WM, WN, 1, PT, TR, TC, 4>(a, ph.p[0], lds); break;
        case 1: conv_mfma_body<2, 3, 1, CIN, WM, WN, 1, PT, TR, TC, 3>(a, ph.p[1], lds); break;
        case 2: conv_mfma_body<3, 2, 1, CIN, WM, WN, 1, PT, TR, TC, 3>(a, ph.p[2], lds); break;
        default: conv_mfma_body<3, 3, 1, CIN, WM, WN, 1, PT, TR, TC, 3>(a, ph.p[3], lds); break;
    }
}

// The same layer with the two x-phases of an output row pair computed by ONE work-group, one after the other on the same
// pixel tile (blockIdx.z = py): lane j then holds the outputs (2 gy + py, 2 gx) and (2 gy + py, 2 gx + 1) of its pixel and
// writes them as one 8-byte store -- consecutive lanes fill whole cache lines.  With a work-group per phase (the kernel
// above) every store instruction writes every other float of its lines and the two halves of a line come from different
// work-groups at different times: rocprofv3 counted 46.7 MB written for a 25.2 MB output (profiles/r03_counters.txt).
typedef float f32x2g __attribute__((ext_vector_type(2)));
template <int CIN, int WM, int WN, int PT, int TR, int TC, int CPC>
__global__ __launch_bounds__(256) void deconv5_pair_kernel(const GArgs a, const GPhases4 ph) {
    __shared__ float lds[GGeo<3, 3, 1, 1, TR, TC, CPC>::LDSF];
    f32x16 v0[PT], v1[PT];
    const int py = blockIdx.z;
    if (py == 0) {               // taps 2x2 then 2x3
        conv_mfma_body<2, 2, 1, CIN, WM, WN, 1, PT, TR, TC, 4, true, false, CPC>(a, ph.p[0], lds, v0);
        conv_mfma_body<2, 3, 1, CIN, WM, WN, 1, PT, TR, TC, 3, true, false, CPC>(a, ph.p[1], lds, v1);
    } else {                     // taps 3x2 then 3x3
        conv_mfma_body<3, 2, 1, CIN, WM, WN, 1, PT, TR, TC, 3, true, false, CPC>(a, ph.p[2], lds, v0);
        conv_mfma_body<3, 3, 1, CIN, WM, WN, 1, PT, TR, TC, 3, true, false, CPC>(a, ph.p[3], lds, v1);
    }
    // same decomposition as the body's: tile, channel tile, pixel of this lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave % WM, wn = (wave / WM) % WN;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx_ = b % a.tiles_x; b /= a.tiles_x;
    const int ty_ = b % a.tiles_y; const int n = b / a.tiles_y;
    const int cot = blockIdx.y * WM + wm;
    const int j = lane & 31, kh = lane >> 5;
    const size_t OHW = (size_t)a.OH * a.OW;
    float* __restrict__ y = a.y + ((size_t)n * a.Cout + 32 * cot + 4 * kh) * OHW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * (PT * wn + p) + j;
        const int gy = ty_ * TR + q / TC, gx = tx_ * TC + q % TC;
        if (gy < a.GH && gx < a.GW) {
            const size_t pix = (size_t)(gy * 2 + py) * a.OW + gx * 2;           // OW = 2 GW: 8-byte aligned
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int crow = (r & 3) + 8 * (r >> 2);
                if (32 * cot + crow + 4 * kh < a.Cout)
                    *(f32x2g*)(y + pix + (size_t)crow * OHW) = f32x2g{v0[p][r], v1[p][r]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static unsigned long long* g_cm_prof = nullptr;
#ifdef CM_PROF      // tuning builds only (a -DCM_PROF build): device buffer of 4 x u64 shader-clock stamps per wave
extern "C" void ic_conv2d_mfma_set_prof(unsigned lo, unsigned hi) { g_cm_prof = (unsigned long long*)(((unsigned long long)hi << 32) | lo); }
#endif
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
    // 3x3 / 2 conv 128 -> <= 128: the data gradient of from_bn (adjoint of its 3x3 / 2 transposed conv) in training
    if (KH == 3 && KW == 3 && stride == 2 && !transposed) return Cin == 128 && Cout <= 128;
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
    a.N = N; a.IH = H; a.IW = W; a.Cout = Cout; a.relu = relu; a.prof = g_cm_prof; a.ncot = ncot;
    if (!transposed) {
        GPhase ph{};
        ph.wp = w_packed; ph.py = 0; ph.px = 0;
        ph.oy0 = -ic_same_pad_before(H, KH, 2); ph.ox0 = -ic_same_pad_before(W, KW, 2);
        a.GH = ic_cdiv(H, 2); a.GW = ic_cdiv(W, 2); a.OH = a.GH; a.OW = a.GW; a.OS = 1;
        if (KH == 3) {              // from_bn's adjoint: as to_bn, 9 taps
            a.tiles_x = ic_cdiv(a.GW, 16); a.tiles_y = ic_cdiv(a.GH, 2);
            hipLaunchKernelGGL((conv_mfma_kernel<3, 3, 2, 128, 1, 1, 4, 1, 2, 16, 3>),
                               dim3(a.tiles_x * a.tiles_y * N, ic_cdiv(Cout, 32)), dim3(256), 0, st, a, ph);   // live channel tiles only
        } else if (Cin == 64) {     // h2: 4 channel tiles x 32 pixels per work-group
            a.tiles_x = ic_cdiv(a.GW, 16); a.tiles_y = ic_cdiv(a.GH, 2);
#ifndef CM_CPC_H2
#define CM_CPC_H2 2         // 8-channel groups per staged chunk (tuning builds: 1 = the round-2 form)
#endif
            hipLaunchKernelGGL((conv_mfma_kernel<5, 5, 2, 64, 4, 1, 1, 1, 2, 16, 5, CM_CPC_H2>),
                               dim3(a.tiles_x * a.tiles_y * N, ncot / 4), dim3(256), 0, st, a, ph);
        } else {                    // to_bn: 1 channel tile x 32 pixels x 4 K-slices per work-group
            a.tiles_x = ic_cdiv(a.GW, 16); a.tiles_y = ic_cdiv(a.GH, 2);
            if (Cout % 32 == 1 && Cout > 1)     // C + 1 channels with the importance map: the odd one rides on the vector unit
                hipLaunchKernelGGL((conv_mfma_x1_kernel<5, 5, 2, 128, 1, 1, 4, 1, 2, 16, 5>),
                                   dim3(a.tiles_x * a.tiles_y * N, (Cout - 1) / 32), dim3(256), 0, st, a, ph);
            else
                hipLaunchKernelGGL((conv_mfma_kernel<5, 5, 2, 128, 1, 1, 4, 1, 2, 16, 5>),
                                   dim3(a.tiles_x * a.tiles_y * N, ic_cdiv(Cout, 32)), dim3(256), 0, st, a, ph);
        }
    } else {                        // h12: four phases, 2 channel tiles x 2 pixel groups of 32 per work-group
        a.GH = H; a.GW = W; a.OH = 2 * H; a.OW = 2 * W; a.OS = 2;
        const int pad = 1, nch = Cin / GKC;
        GPhases4 ph{};
        size_t off = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                int nty, ntx, kyb, kxb;
                GPhase& g = ph.p[py * 2 + px];
                phase_taps(KH, pad, py, &nty, &g.oy0, &kyb);
                phase_taps(KW, pad, px, &ntx, &g.ox0, &kxb);
                g.wp = w_packed + off; g.py = py; g.px = px;
                off += (size_t)nch * nty * ntx * ncot * 256;
            }
        a.tiles_x = ic_cdiv(a.GW, 16); a.tiles_y = ic_cdiv(a.GH, 4);
#ifdef CM_UNPAIRED      // tuning builds: the round-2 form, a work-group per phase
        hipLaunchKernelGGL((deconv5_mfma_kernel<128, 2, 2, 1, 4, 16>),
                           dim3(a.tiles_x * a.tiles_y * N, ncot / 2, 4), dim3(256), 0, st, a, ph);
#else
#ifndef CM_CPC_H12
#define CM_CPC_H12 2
#endif
        hipLaunchKernelGGL((deconv5_pair_kernel<128, 2, 2, 1, 4, 16, CM_CPC_H12>),
                           dim3(a.tiles_x * a.tiles_y * N, ncot / 2, 2), dim3(256), 0, st, a, ph);
#endif
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}
