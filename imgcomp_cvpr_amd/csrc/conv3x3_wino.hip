// 3x3, stride 1, 128 -> 128 channel convolution of the residual stacks (autoencoder.py:224-234, :252-262: 32 of the 35
// encoder convs and 32 of the 35 decoder layers) in Winograd F(2x2, 3x3) form on the fp32 matrix cores.
//
//   Y = At [ (G g Gt) (.) (Bt d B) ] A          g: 3x3 filter, d: 4x4 input patch, Y: 2x2 outputs ("tile")
//
// The channel contraction of the 16 transform positions is 16 independent GEMMs  M_p[co][tile] = sum_ci U_p[co][ci] V_p[ci][tile]
// -- 16 / 36 of the multiply-adds of the direct form.  How it is laid onto a gfx950 wave:
//
//   * v_mfma_f32_32x32x2_f32 computes D[32 x 32] += A[32 x 2] B[2 x 32]; lane l supplies A[l & 31][l >> 5] and
//     B[l >> 5][l & 31].  M axis = 32 output channels, N axis = 32 tiles, K = 2 input channels.
//   * the B operand of position p is V_p[ci = 2 ks + (l >> 5)][tile = l & 31]: ONE input patch transform per lane yields
//     the B operands of all 16 positions of that k-step.  Each lane loads the aligned pixel pair of its tile for the 4
//     patch rows (coalesced dwordx2) and takes the two outer columns from its neighbour lanes by DPP row shifts, does
//     the 32 add/subs of Bt d B in registers and feeds 16 MFMAs.
//   * the A operands (transformed filters) are pre-packed in fragment order: 4 global_load_dwordx4 per k-step, each
//     reading 1 KB contiguous across the wave (L2 resident, 1 MB per layer).
//   * a wave owns (32 output channels) x (32 tiles) x (16 positions) = 16 accumulators of 16 registers = 256 AGPRs;
//     the output transform At M A is then register-local (same lane, same register index across the 16 accumulators),
//     followed by BN scale/shift, ReLU, residual adds and 8-byte stores (a tile row is 2 neighbouring pixels).
//   * one wave per SIMD (512 registers): latency is hidden by a 4-stage register ring -- patch and filter fragments
//     are requested 3 k-steps (~3000 clocks of MFMA work) before they are consumed.
//   * work-group = 4 waves = the 4 output-channel tiles of the same 32 spatial tiles.  In the per-wave form every wave
//     loads and transforms every k-step (the patch loads of the four waves hit the same L1 lines); in the SHARED form
//     (wino3x3_c128_shared_kernel, the one the even-width maps run) wave w loads and transforms only k-steps 4j + w and
//     the B operands travel to the other waves through a 32 KB LDS ring, one barrier per 4 k-steps -- a quarter of the
//     vector instructions per MFMA.  The filter streams are disjoint.
//   * launches: full rounds of 256 tile groups run this whole-K form; small maps and small remainders run the K-split
//     form (one channel tile per work-group, the four waves = four quarters of the input channels), see wino_plan.
//
// Rounding differs from the direct form (different summation tree); measured against the float64 oracle both stay
// inside the 1e-4 parity bound (tests/test_gpu_ops.py).
#include "wino_common.h"
#include <type_traits>
#include <cstdlib>
#include "internal.h"

#define WN_STAGES 4
#ifndef WN_ABL
#define WN_ABL 0      // tuning builds only: 1 no patch loads, 2 no filter loads, 4 no transform in the main loop
#endif

// ---- filter transform + packing -----------------------------------------------------------------------------------
// U = G g Gt, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// packed index: ((((cot * 64 + ks) * 4 + pos / 4) * 64 + lane) * 4 + pos % 4), cot = co / 32, ks = ci / 2,
// lane = (ci & 1) * 32 + (co & 31): each of the 4 dwordx4 loads of a k-step reads 1 KB contiguous across the wave
// (a 64-byte lane stride costs the L1 one line per lane and made the kernel L1-bound at 2200 clocks per k-step).
// backward = 1 packs the adjoint (data-gradient) filter: g'[a][b][in = co][out = ci] = g[2 - a][2 - b][ci][co].
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w_tf, float* __restrict__ out, int backward) {
    const int idx = blockIdx.x * 256 + threadIdx.x;        // (in, out) pair
    if (idx >= WN_C * WN_C) return;
    const int cin = idx / WN_C, cout = idx % WN_C;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = backward ? w_tf[(((2 - a) * 3 + (2 - b)) * WN_C + cout) * WN_C + cin]
                               : w_tf[((a * 3 + b) * WN_C + cin) * WN_C + cout];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    float* o = out + (((size_t)(cout >> 5) * 64 + (cin >> 1)) * 4 * 64 + ((cin & 1) * 32 + (cout & 31))) * 4;
    // second layout, for v_mfma_f32_16x16x4_f32 (wino3x3_c128_t16_kernel): cot16 = co / 16, ks4 = ci / 4,
    // lane = (ci & 3) * 16 + (co & 15)
    float* o16 = out + WN_FRAG_FLOATS + (((size_t)(cout >> 4) * 32 + (cin >> 2)) * 4 * 64 + ((cin & 3) * 16 + (cout & 15))) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        f32x4 q = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        *(f32x4*)(o + a * 256) = q;
        *(f32x4*)(o16 + a * 256) = q;
    }
}

// ---- the convolution ----------------------------------------------------------------------------------------------
// A wave owns 2 x 16 tiles (4 x 32 output pixels): a tile row is 16 lanes = one DPP row (helpers in wino_common.h).
// VEC: even W, the pixel pair (2tx, 2tx+1) is one aligned 8-byte load.
// KS = 1: the four waves of a work-group are the four output-channel tiles of one tile group (64 k-steps each).
// KS = 4: small maps that cannot fill the chip with 4 x groups waves -- a work-group is ONE channel tile (cot) of a
//         group, its four waves take a quarter of the input channels each (16 k-steps) and the partial accumulators are
//         summed through LDS in the fixed order (w0 + w1) + (w2 + w3); wave w then finishes output registers 4w..4w+3.
template <bool VEC, int KS, bool SHARE = false, bool WT = false>
__device__ __forceinline__ void wino_body(const WnArgs& a, int n, int gy, int gx, int cot_in) {
#ifdef WN_PROF
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform (scalar) on purpose
    const int cot = KS == 1 ? wave : cot_in;                                // this wave's output-channel tile
    constexpr int NKS = 64 / KS;                                            // k-steps per wave
    const int ks0 = KS == 1 ? 0 : wave * NKS;
    const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int txl = li & 15;
    const int ty = gy * 2 + (li >> 4), tx = gx * 16 + txl;
    const int r0 = 2 * ty - 1;
    const int H = a.H, W = a.W;
    const int HW = H * W;

    // All loads are raw buffer loads: descriptor + loop-invariant per-lane byte offset + per-k-step SCALAR offset, so
    // the main loop has no vector address arithmetic; a padded position (row/column outside the image, tile outside
    // the map) gets an out-of-range lane offset and the hardware returns 0 -- zero padding without a select, a clamp or
    // a branch, and border work-groups run the same instruction stream as interior ones.
    //
    // Patch columns 2tx-1 .. 2tx+2: every lane loads its own aligned pair (2tx, 2tx+1); the outer two columns are the
    // neighbour lanes' values (DPP row shift) except at the two ends of the 16-lane tile row, which load one extra
    // dword per patch row.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.x + (size_t)n * WN_C * HW), 0, WN_C * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, WN_PACKED_FLOATS * 4, 0x00020000);
    // VEC: the end-of-row value is fetched as the aligned pair that contains it -- lane 0: (2tx-2, 2tx-1), lane 15:
    // (2tx+2, 2tx+3) -- so that each half is the 'old' operand of one DPP move and is overwritten in place (no copies).
    const int ecol = VEC ? (txl == 0 ? 2 * tx - 2 : (txl == 15 ? 2 * tx + 2 : -1))
                         : (txl == 0 ? 2 * tx - 1 : (txl == 15 ? 2 * tx + 2 : -1));
    unsigned o0[4], o1[4], oe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        const bool rok = r >= 0 && r < H;
        const unsigned rb = (unsigned)(kh * HW + r * W) * 4u;
        o0[i] = (rok && 2 * tx < W) ? rb + 8u * tx : WN_OOB;
        o1[i] = (rok && 2 * tx + 1 < W) ? rb + 8u * tx + 4u : WN_OOB;
        oe[i] = (rok && ecol >= 0 && ecol < W) ? rb + 4u * ecol : WN_OOB;
    }
    const unsigned fo = lane * 16u;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x2 pp[WN_STAGES][4];      // own pair of the 4 patch rows
    f32x2 pe[WN_STAGES][4];      // end-of-row pair (only meaningful in lanes 0 / 15 of a tile row); !VEC: [0] only
    f32x4 fl[WN_STAGES][4];

    auto load_patch = [&](int s, int ks) {
        const int so = ks * 2 * HW * 4;                     // scalar: channel pair 2 ks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (VEC) {
                pp[s][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, o0[i], so, 0));
                pe[s][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, oe[i], so, 0));
            } else {
                pp[s][i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o0[i], so, 0));
                pp[s][i][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o1[i], so, 0));
                pe[s][i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, oe[i], so, 0));
            }
        }
    };
    auto load_filter = [&](int s, int ks) {
        const int so = (cot * 64 + ks) * 4096;              // scalar: 4 KB per (channel tile, k-step)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            fl[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + q * 1024u, so, 0));
    };

    if constexpr (!SHARE) {
    // (pinned in ring order: the wait counts at the loop head are the merge of this entry state and the back edge; a
        // prologue load scheduled late would make every iteration wait for almost everything in flight)
    #pragma unroll
        for (int s = 0; s < WN_STAGES - 1; ++s) {
            load_patch(s, ks0 + s);
            __builtin_amdgcn_sched_barrier(0);
            load_filter(s, ks0 + s);
            __builtin_amdgcn_sched_barrier(0);
        }
    
}

    // Bt d B of one lane's patch -> the 16 B operands of a k-step.  Written on (x0, x1) / (left, right) pairs so that
    // it maps onto packed fp32 adds: every vector instruction costs matrix-pipe time when there is one wave per SIMD.
    auto transform = [&](int s, float (&v)[16]) {
        f32x2 A[4], B[4];       // A = (x0, x1);  B = (right, left)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[i] = pp[s][i];
            if (VEC) {
                B[i][0] = dpp_from_right(pe[s][i][0], A[i][0]);
                B[i][1] = dpp_from_left(pe[s][i][1], A[i][1]);
            } else {
                B[i][0] = dpp_from_right(pe[s][i][0], A[i][0]);
                B[i][1] = dpp_from_left(pe[s][i][0], A[i][1]);
            }
        }
        // rows: u0 = d0 - d2, u1 = d1 + d2, u2 = d2 - d1, u3 = d1 - d3 (on column pairs)
        const f32x2 uA[4] = {pk_sub(A[0], A[2]), pk_add(A[1], A[2]), pk_sub(A[2], A[1]), pk_sub(A[1], A[3])};
        const f32x2 uB[4] = {pk_sub(B[0], B[2]), pk_add(B[1], B[2]), pk_sub(B[2], B[1]), pk_sub(B[1], B[3])};
        // columns (c0, c1, c2, c3) = (B.y, A.x, A.y, B.x): v0 = c0 - c2, v1 = c1 + c2, v2 = c2 - c1, v3 = c1 - c3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 v30, v12;
            // (v3, v0) = (A.x - B.x, B.y - A.y)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v30) : "v"(uA[i]), "v"(uB[i]));
            // (v1, v2) = (A.x + A.y, A.y - A.x)
            asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(uA[i]));
            v[4 * i] = v30[1]; v[4 * i + 1] = v12[0]; v[4 * i + 2] = v12[1]; v[4 * i + 3] = v30[0];
        }
    };

#ifdef WN_PROF
    unsigned long long pd1 = 0, pd2 = 0, pd3 = 0, pt0 = 0;
#endif
    if constexpr (SHARE) {
        // The four waves of a whole-K work-group multiply the SAME transformed input (B operand) with four different
        // channel tiles of the filter.  Instead of every wave loading and transforming every k-step, wave w produces
        // k-steps 4j + w and the B operands travel through a ring in LDS: a quarter of the patch loads and transform
        // instructions per wave -- vector instructions that, with one wave per SIMD, come straight out of the matrix
        // pipe's issue time.  One iteration = 4 k-steps: ring half (j & 1) is read while half ((j+1) & 1) is written,
        // one barrier per iteration.
        static_assert(KS == 1, "shared input transform is the whole-K form");
        __shared__ f32x4 ring[2 * 4 * 4 * 64];                // [half][k-step of the iteration][position quad][lane]
        auto put = [&](int half, const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = {vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]};
                ring[((half * 4 + wave) * 4 + q) * 64 + lane] = t;
            }
        };
        auto get = [&](int half, int st, f32x4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = ring[((half * 4 + st) * 4 + q) * 64 + lane];
        };
        load_patch(0, wave);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < WN_STAGES - 1; ++st) {
            load_filter(st, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_patch(1, 4 + wave);
        __builtin_amdgcn_sched_barrier(0);
        float vt[16];
        transform(0, vt);
        put(0, vt);
        __syncthreads();
        f32x4 bq[2][4];
        get(0, 0, bq[0]);
#ifdef WN_PROF
        pt0 = __builtin_amdgcn_s_memtime();
#endif
        for (int jj = 0; jj < 16; jj += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                     // iteration j = jj + u reads ring half u
                const int j = jj + u;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int ks = 4 * j + st;
                    load_filter((st + WN_STAGES - 1) % WN_STAGES, ks + WN_STAGES - 1 < 64 ? ks + WN_STAGES - 1 : 63);
#ifndef WN_PATCH_ST
#define WN_PATCH_ST 0
#endif
                    if (st == WN_PATCH_ST) {
                        // own k-step of iteration j + 2 into the patch slot transformed one iteration ago.  Past the end it
                        // repeats the last own k-step: same instruction stream, the extra ring entries are never read.
                        const int kp = 4 * (j + 2) + wave < 64 ? 4 * (j + 2) + wave : 60 + wave;
                        load_patch(u, kp);
                    }
                    // own k-step of iteration j + 1 (loaded one iteration ago) through the transform
                    if (st == 0) transform(u ^ 1, vt);
                    if (st == 1) put(u ^ 1, vt);
                    if (st < 3) get(u, st + 1, bq[(st + 1) & 1]);
                    else get(u ^ 1, 0, bq[0]);
#pragma unroll
                    for (int p = 0; p < 16; ++p)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fl[st][p >> 2][p & 3], bq[st & 1][p >> 2][p & 3], acc[p], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             // 1 MFMA
                        if (p < (st == WN_PATCH_ST ? 12 : 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                        if (p >= 4 && p < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 LDS read
                        if (st == 1 && p >= 8 && p < 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 LDS write
                        if (st == 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                // the transform
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (st == 2) {
                        // every wave has written its k-step of the next half and read all of this one
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
#ifdef WN_PROF
        pd1 = __builtin_amdgcn_s_memtime() - pt0; pt0 += pd1;
#endif
    } else {
    float v[2][16];
    transform(0, v[0]);
#ifdef WN_PROF
    pt0 = __builtin_amdgcn_s_memtime();
#endif
    for (int k0 = 0; k0 < NKS; k0 += WN_STAGES) {
#pragma unroll
        for (int s = 0; s < WN_STAGES; ++s) {
            const int ks = ks0 + k0 + s;
            const int cur = s & 1, nxt = cur ^ 1;
            // Request k-step ks + 3 into the ring slot whose filter was consumed one k-step ago and whose patch was
            // transformed two k-steps ago.  Always issued (past the end it re-reads the last k-step) so the loop body
            // is ONE basic block and outstanding loads are counted exactly.
            const int sn = (s + WN_STAGES - 1) % WN_STAGES;
            const int kn = ks + WN_STAGES - 1 < ks0 + NKS ? ks + WN_STAGES - 1 : ks0 + NKS - 1;
            if (!(WN_ABL & 1)) load_patch(sn, kn);
            if (!(WN_ABL & 2)) load_filter(sn, kn);
            // the input transform of the NEXT k-step
            transform((s + 1) % WN_STAGES, v[nxt]);
#pragma unroll
            for (int p = 0; p < 16; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fl[s][p >> 2][p & 3], v[cur][p], acc[p], 0, 0, 0);
            // Issue order inside the k-step: one memory request and a few transform instructions behind every MFMA.
            // With one wave per SIMD nothing else can fill the issue slot, so whatever is not tucked behind a running
            // MFMA is exposed; left alone, the scheduler sinks the loads next to their uses (to cut register pressure)
            // and turns the ring into load-wait-use.
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // 1 MFMA
                if (p < (VEC ? 12 : 16)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // a little VALU
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef WN_PROF
            const unsigned long long q3 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            pd1 += q3 - pt0; pt0 = q3;
#endif
        }
    }
    }
#ifdef WN_PROF
    // stamps: entry -> loop start (prologue), loop, loop end -> last store issued (epilogue); written at the very end
    const unsigned long long t_loop0 = pt0 - pd1, t_loop1 = pt0;
#endif

    // ---- At M A, BN fold, activation, residuals, store ----
    const int oy = 2 * ty, ox = 2 * tx;
    const bool inside = oy < H && ox < W;
    const bool row1 = oy + 1 < H, col1 = ox + 1 < W;
    const bool vec = col1 && ((W & 1) == 0);
    const long long obase = (long long)n * WN_C * HW + (long long)oy * W + ox;
    // BN fold, activation, residuals and store of one output channel of this lane's tile (any width)
    auto finish = [&](int co, float o00, float o01, float o10, float o11) __attribute__((always_inline)) {
        const float sc = a.scale[co], sh = a.shift[co];
        o00 = fmaf(o00, sc, sh); o01 = fmaf(o01, sc, sh); o10 = fmaf(o10, sc, sh); o11 = fmaf(o11, sc, sh);
        if (a.relu) { o00 = fmaxf(o00, 0.f); o01 = fmaxf(o01, 0.f); o10 = fmaxf(o10, 0.f); o11 = fmaxf(o11, 0.f); }
        const long long o = obase + (long long)co * HW;
        if (vec) {
            f32x2 q0 = {o00, o01}, q1 = {o10, o11};
            if (a.res1) {
                const f32x2 e0 = *(const f32x2*)(a.res1 + o);
                q0 += e0;
                if (row1) q1 += *(const f32x2*)(a.res1 + o + W);
            }
            if (a.res2) {
                const f32x2 e0 = *(const f32x2*)(a.res2 + o);
                q0 += e0;
                if (row1) q1 += *(const f32x2*)(a.res2 + o + W);
            }
            *(f32x2*)(a.y + o) = q0;
            if (row1) *(f32x2*)(a.y + o + W) = q1;
        } else {
            if (a.res1) {
                o00 += a.res1[o];
                if (col1) o01 += a.res1[o + 1];
                if (row1) { o10 += a.res1[o + W]; if (col1) o11 += a.res1[o + W + 1]; }
            }
            if (a.res2) {
                o00 += a.res2[o];
                if (col1) o01 += a.res2[o + 1];
                if (row1) { o10 += a.res2[o + W]; if (col1) o11 += a.res2[o + W + 1]; }
            }
            a.y[o] = o00;
            if (col1) a.y[o + 1] = o01;
            if (row1) { a.y[o + W] = o10; if (col1) a.y[o + W + 1] = o11; }
        }
    };
    // one output channel of this lane's tile from its 16 position sums
    auto emit = [&](int co, const float (&m)[16]) __attribute__((always_inline)) {
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = m[j] + m[4 + j] + m[8 + j];
            t1[j] = m[4 + j] - m[8 + j] - m[12 + j];
        }
        finish(co, t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3], t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]);
    };
    if (KS == 1) {
        if (!VEC && !inside) return;
        if constexpr (VEC) {                               // even W: every inside tile has both columns, pairs are aligned
            // Branch-free: residuals and the output go through buffer descriptors.  An absent residual is a descriptor
            // of zero records (every load returns 0), a tile row below the map or a tile outside it an out-of-range lane
            // offset (loads return 0, stores are dropped): no per-channel control flow for the register allocator to
            // fight, border work-groups run the interior instruction stream.
            // The operands of channel r + 5 (BN scale/shift, residual rows) are requested while channel r is finished:
            // fetched inside each iteration, every one of the 16 waited out its own L2 round trip (10.6k of a wave's 93k
            // clocks); all 16 at once would need 160 registers next to the 256 accumulators and spill into the main loop.
            const int img_bytes = WN_C * HW * 4;
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)n * WN_C * HW), 0, img_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.res1 ? a.res1 + (size_t)n * WN_C * HW : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.res2 ? a.res2 + (size_t)n * WN_C * HW : a.x), 0, a.res2 ? img_bytes : 0, 0x00020000);
            const unsigned lo0 = inside ? (unsigned)((4 * kh * HW + oy * W + ox) * 4) : WN_OOB;
            const unsigned lo1 = inside && row1 ? lo0 + 4u * W : WN_OOB;
#ifndef WN_EPD
#define WN_EPD 6          // operand slots of the epilogue pipeline
#define WN_EPA 5          // channels requested ahead
#define WN_EPU 1          // channels finished per scheduling region
#endif
            // Round 3, measured with -DWN_PROF on a Kodak map (8.35 k clocks of epilogue per wave): two or four channels per
            // scheduling region (more independent chains for the one resident wave) change nothing -- 8.62 k / 8.42 k.  The
            // epilogue of a lock-step round is a bandwidth burst: every work-group stores its 64 KB and reads one or two
            // residuals of the same size within the same ~3.5 us (25-38 MB: 7-11 TB/s demanded), not an issue-bound loop.
            constexpr int EPD = WN_EPD, EPA = WN_EPA, EPU = WN_EPU;
            static_assert(EPA + EPU <= EPD && 16 % EPU == 0, "a slot is refilled only after its channel has been finished");
            float sc[EPD], sh[EPD];
            f32x2 ra0[EPD], ra1[EPD], rb0[EPD], rb1[EPD];
            const float relu_lo = a.relu ? 0.f : -__builtin_inff();
            const float* __restrict__ scp = a.scale + 32 * cot + 4 * kh;
            const float* __restrict__ shp = a.shift + 32 * cot + 4 * kh;
            auto fetch = [&](int r) __attribute__((always_inline)) {
                const int cr = (r & 3) + 8 * (r >> 2), sl = r % EPD;              // channel = 32 cot + 4 kh + cr
                sc[sl] = scp[cr]; sh[sl] = shp[cr];
                const int so = (32 * cot + cr) * HW * 4;                          // scalar part of the channel offset
                ra0[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo0, so, 0));
                ra1[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo1, so, 0));
                rb0[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo0, so, 0));
                rb1[sl] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo1, so, 0));
            };
#pragma unroll
            for (int r = 0; r < EPA; ++r) fetch(r);
#pragma unroll
            for (int r0_ = 0; r0_ < 16; r0_ += EPU) {
#pragma unroll
                for (int u = 0; u < EPU; ++u)
                    if (r0_ + u + EPA < 16) fetch(r0_ + u + EPA);
#pragma unroll
                for (int u = 0; u < EPU; ++u) {
                    const int r = r0_ + u;
                    const int cr = (r & 3) + 8 * (r >> 2), sl = r % EPD;
                    float t0[4], t1[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m0 = acc[j][r], m1 = acc[4 + j][r], m2 = acc[8 + j][r], m3 = acc[12 + j][r];
                        t0[j] = m0 + m1 + m2;
                        t1[j] = m1 - m2 - m3;
                    }
                    float o00 = t0[0] + t0[1] + t0[2], o01 = t0[1] - t0[2] - t0[3];
                    float o10 = t1[0] + t1[1] + t1[2], o11 = t1[1] - t1[2] - t1[3];
                    o00 = fmaf(o00, sc[sl], sh[sl]); o01 = fmaf(o01, sc[sl], sh[sl]);
                    o10 = fmaf(o10, sc[sl], sh[sl]); o11 = fmaf(o11, sc[sl], sh[sl]);
                    o00 = fmaxf(o00, relu_lo); o01 = fmaxf(o01, relu_lo); o10 = fmaxf(o10, relu_lo); o11 = fmaxf(o11, relu_lo);
                    f32x2 q0 = {o00, o01}, q1 = {o10, o11};
                    q0 += ra0[sl]; q1 += ra1[sl];
                    q0 += rb0[sl]; q1 += rb1[sl];
                    const int so = (32 * cot + cr) * HW * 4;
                    // WT: a launch that is one round of work-groups stores write-through (sc1): nothing is left dirty in the L2s
                    // for the kernel boundary to write back (measured on the NB-segment kernel: 1 us per layer)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q0), yr, lo0, so, WT ? 16 : 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q1), yr, lo1, so, WT ? 16 : 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float m[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) m[p] = acc[p][r];
                emit(32 * cot + 8 * (r >> 2) + 4 * kh + (r & 3), m);
            }
        }
#ifdef WN_PROF
        if (a.prof && lane == 0) {
            unsigned long long* d = a.prof + 4 * ((size_t)blockIdx.x * 4 + wave);
            d[0] = t_loop0 - t_entry; d[1] = t_loop1 - t_loop0; d[2] = __builtin_amdgcn_s_memtime() - t_loop1; d[3] = t_entry;
        }
#endif
    } else {
        // cross-wave sum AFTER the output transform: At M A is linear, so every wave first reduces its 16 partial position
        // sums to the 4 outputs of each (channel, tile) and the waves exchange a quarter of the accumulator volume -- one
        // phase and one barrier through 64 KB of LDS instead of eight phases in the Winograd domain.  Wave w then finishes
        // output registers 4w..4w+3 (channels 32 cot + 8w + 4kh + i); sum order (w0 + w1) + (w2 + w3).
        __shared__ f32x4 red[4 * 4 * 4 * 64];                 // [source wave][output][register quad][lane]
        f32x16 out[4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = acc[j][r], m1 = acc[4 + j][r], m2 = acc[8 + j][r], m3 = acc[12 + j][r];
                t0[j] = m0 + m1 + m2;
                t1[j] = m1 - m2 - m3;
            }
            out[0][r] = t0[0] + t0[1] + t0[2]; out[1][r] = t0[1] - t0[2] - t0[3];
            out[2][r] = t1[0] + t1[1] + t1[2]; out[3][r] = t1[1] - t1[2] - t1[3];
        }
        // epilogue operands of this wave's four channels, requested before the exchange (the accumulators are dead now)
        float esc[4], esh[4];
        f32x2 ea0[4], ea1[4], eb0[4], eb1[4];
        const long long eo0 = obase + (long long)(32 * cot + 8 * wave + 4 * kh) * HW;     // channel = 32 cot + 8 wave + 4 kh + i
        if (VEC && inside) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 32 * cot + 8 * wave + 4 * kh + i;
                esc[i] = a.scale[co]; esh[i] = a.shift[co];
                const long long o = eo0 + (long long)i * HW;
                if (a.res1) { ea0[i] = *(const f32x2*)(a.res1 + o); ea1[i] = row1 ? *(const f32x2*)(a.res1 + o + W) : f32x2{0.f, 0.f}; }
                if (a.res2) { eb0[i] = *(const f32x2*)(a.res2 + o); eb1[i] = row1 ? *(const f32x2*)(a.res2 + o + W) : f32x2{0.f, 0.f}; }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 q = {out[o][4 * r4], out[o][4 * r4 + 1], out[o][4 * r4 + 2], out[o][4 * r4 + 3]};
                red[((wave * 4 + o) * 4 + r4) * 64 + lane] = q;
            }
        __syncthreads();
        f32x4 mine[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const f32x4* q = red + (o * 4 + wave) * 64 + lane;
            mine[o] = (q[0] + q[16 * 64]) + (q[2 * 16 * 64] + q[3 * 16 * 64]);
        }
#ifdef WN_PROF
        const unsigned long long t_xchg = __builtin_amdgcn_s_memtime();
#endif
        if (inside) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o00 = mine[0][i], o01 = mine[1][i], o10 = mine[2][i], o11 = mine[3][i];
            if constexpr (VEC) {
                o00 = fmaf(o00, esc[i], esh[i]); o01 = fmaf(o01, esc[i], esh[i]);
                o10 = fmaf(o10, esc[i], esh[i]); o11 = fmaf(o11, esc[i], esh[i]);
                if (a.relu) { o00 = fmaxf(o00, 0.f); o01 = fmaxf(o01, 0.f); o10 = fmaxf(o10, 0.f); o11 = fmaxf(o11, 0.f); }
                f32x2 q0 = {o00, o01}, q1 = {o10, o11};
                if (a.res1) { q0 += ea0[i]; q1 += ea1[i]; }
                if (a.res2) { q0 += eb0[i]; q1 += eb1[i]; }
                const long long o = eo0 + (long long)i * HW;
                *(f32x2*)(a.y + o) = q0;
                if (row1) *(f32x2*)(a.y + o + W) = q1;
            } else {
                const int r = 4 * wave + i;                // register r of the tile: channel 8 (r >> 2) + 4 kh + (r & 3)
                finish(32 * cot + 8 * (r >> 2) + 4 * kh + (r & 3), o00, o01, o10, o11);
            }
        }
        }
#ifdef WN_PROF
        if (a.prof && lane == 0) {     // K-split: prologue, k-loop, transform + exchange, rest of the epilogue
            unsigned long long* d = a.prof + 4 * ((size_t)blockIdx.x * 4 + wave);
            d[0] = t_loop0 - t_entry; d[1] = t_loop1 - t_loop0; d[2] = t_xchg - t_loop1; d[3] = __builtin_amdgcn_s_memtime() - t_xchg;
        }
#endif
    }
}

// VEC = even W (aligned pixel pairs); odd widths get their own kernel.
template <bool VEC>
__global__ __launch_bounds__(256) void wino3x3_c128_kernel(const WnArgs a) {
    const int gx = blockIdx.x % a.gcols;
    const int t = blockIdx.x / a.gcols;
    wino_body<VEC, 1>(a, t / a.grows, t % a.grows, gx, 0);
}

// whole-K with the input transform shared between the four waves through LDS (even widths)
template <bool WT>
__global__ __launch_bounds__(256) void wino3x3_c128_shared_kernel(const WnArgs a) {
    // work-group i runs on XCD i % 8 (round-robin dispatch): give every XCD a contiguous run of tile groups, so that the
    // halo rows shared by vertically adjacent groups are found in ONE L2 instead of being fetched by several
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int gx = b % a.gcols;
    const int t = b / a.gcols;
    wino_body<true, 1, true, WT>(a, t / a.grows, t % a.grows, gx, 0);
}

// K-split form for maps that do not fill the chip: one work-group per (tile group, channel tile)
template <bool VEC>
__global__ __launch_bounds__(256) void wino3x3_c128_ksplit_kernel(const WnArgs a) {
    // the four channel tiles of a tile group read the same input: keep them (and the neighbouring groups) on one XCD
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int cot = b & 3;
    const int g = (b >> 2) + a.g0;
    const int gx = g % a.gcols;
    const int t = g / a.gcols;
    wino_body<VEC, 4>(a, t / a.grows, t % a.grows, gx, cot);
}

#ifdef IC_TUNING        // forms that lost every measurement and that the automatic plan never picks: tuning builds only (make TUNING=1)
// ---- 16 tiles x 16 channels per wave on v_mfma_f32_16x16x4_f32 -------------------------------------------------------------
// The 32 x 32 form above makes (tile groups x 4) wave-jobs of 1024 MFMAs; a Kodak map is 768 of them for 1024 SIMDs and a
// quarter of the chip idles.  Here a wave-job is 16 tiles (ONE tile row of 16 = 2 x 32 output pixels) x 16 output channels:
// 16 positions x 4 accumulator registers = 64 instead of 256, four times as many jobs (3072 for a Kodak map = three per
// SIMD), half as long (32 k-steps of 4 input channels, 16 MFMAs of 32 clocks each).  A work-group is one tile row x one HALF of
// the output channels (4 waves = 4 channel tiles); as in the shared kernel, wave w loads and transforms the input of k-steps
// 4j + w only and the B operands reach the other waves through the LDS ring.  ~240 registers per lane: two work-groups per CU.
// Lane l: B operand k = l >> 4 (input channel of the k-step), n = l & 15 (tile); A operand row = l & 15 (channel of the tile),
// k = l >> 4; D rows 4 (l >> 4) .. +3 (channels), column l & 15 (tile).
__global__ __launch_bounds__(256) void wino3x3_c128_t16_kernel(const WnArgs a) {
#ifdef WN_PROF
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    __shared__ f32x4 ring[2 * 4 * 4 * 64];                    // [half][k-step of the iteration][position quad][lane]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kq = lane >> 4, tj = lane & 15;
    // work-groups of this launch: tile groups [g0, g0 + gridDim.x / 4) of the 32 x 32 form, each as 2 tile rows x 2 channel
    // halves; channel half major, so that the work-groups sharing a CU stream the same half of the filter
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int per_half = gridDim.x >> 1;
    const int hc = b >= per_half;
    const int rem = hc ? b - per_half : b;
    const int g = a.g0 + (rem >> 1);
    const int gx = g % a.gcols, t = g / a.gcols;
    const int n = t / a.grows, ty = 2 * (t % a.grows) + (rem & 1);
    if (2 * ty >= a.H) return;                                 // second tile row of a group below the map (whole work-group)
    const int ct = 4 * hc + wave;                              // 16-channel output tile of this wave
    const int tx = gx * 16 + tj;
    const int r0 = 2 * ty - 1;
    const int H = a.H, W = a.W, HW = H * W;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * WN_C * HW), 0, WN_C * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
    const int ecol = tj == 0 ? 2 * tx - 2 : (tj == 15 ? 2 * tx + 2 : -1);
    unsigned o0[4], oe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + i;
        const bool rok = r >= 0 && r < H;
        const unsigned rb = (unsigned)(kq * HW + r * W) * 4u;
        o0[i] = (rok && 2 * tx < W) ? rb + 8u * tx : WN_OOB;
        oe[i] = (rok && ecol >= 0 && ecol < W) ? rb + 4u * ecol : WN_OOB;
    }
    const unsigned fo = lane * 16u;

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[p][r] = 0.f;
    f32x2 pp[2][4], pe[2][4];
    f32x4 fl[WN_STAGES][4];
    auto load_patch = [&](int s, int ks) {
        const int so = ks * 4 * HW * 4;                     // scalar: channels 4 ks ..
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pp[s][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, o0[i], so, 0));
            pe[s][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, oe[i], so, 0));
        }
    };
    auto load_filter = [&](int s, int ks) {
        const int so = (ct * 32 + ks) * 4096;               // scalar: 4 KB per (channel tile, k-step)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            fl[s][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + q * 1024u, so, 0));
    };
    auto transform = [&](int s, float (&v)[16]) {
        f32x2 A[4], B[4];       // A = (x0, x1);  B = (right, left)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A[i] = pp[s][i];
            B[i][0] = dpp_from_right(pe[s][i][0], A[i][0]);
            B[i][1] = dpp_from_left(pe[s][i][1], A[i][1]);
        }
        const f32x2 uA[4] = {pk_sub(A[0], A[2]), pk_add(A[1], A[2]), pk_sub(A[2], A[1]), pk_sub(A[1], A[3])};
        const f32x2 uB[4] = {pk_sub(B[0], B[2]), pk_add(B[1], B[2]), pk_sub(B[2], B[1]), pk_sub(B[1], B[3])};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 v30, v12;
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v30) : "v"(uA[i]), "v"(uB[i]));
            asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(uA[i]));
            v[4 * i] = v30[1]; v[4 * i + 1] = v12[0]; v[4 * i + 2] = v12[1]; v[4 * i + 3] = v30[0];
        }
    };
    auto put = [&](int half, const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 tq = {vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]};
            ring[((half * 4 + wave) * 4 + q) * 64 + lane] = tq;
        }
    };
    auto get = [&](int half, int st, f32x4 (&bb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = ring[((half * 4 + st) * 4 + q) * 64 + lane];
    };
    constexpr int NKS = 32;
    load_patch(0, wave);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < WN_STAGES - 1; ++st) {
        load_filter(st, st);
        __builtin_amdgcn_sched_barrier(0);
    }
    load_patch(1, 4 + wave);
    __builtin_amdgcn_sched_barrier(0);
    float vt[16];
    transform(0, vt);
    put(0, vt);
    __syncthreads();
    f32x4 bq[2][4];
    get(0, 0, bq[0]);
#ifdef WN_PROF
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    for (int jj = 0; jj < NKS / 4; jj += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                         // iteration j = jj + u reads ring half u
            const int j = jj + u;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int ks = 4 * j + st;
                load_filter((st + WN_STAGES - 1) % WN_STAGES, ks + WN_STAGES - 1 < NKS ? ks + WN_STAGES - 1 : NKS - 1);
                if (st == 0) {
                    const int kp = 4 * (j + 2) + wave < NKS ? 4 * (j + 2) + wave : NKS - 4 + wave;
                    load_patch(u, kp);
                    transform(u ^ 1, vt);
                }
                if (st == 1) put(u ^ 1, vt);
                if (st < 3) get(u, st + 1, bq[(st + 1) & 1]);
                else get(u ^ 1, 0, bq[0]);
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(fl[st][p >> 2][p & 3], bq[st & 1][p >> 2][p & 3], acc[p], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                             // 1 MFMA
                    if (p < (st == 0 ? 12 : 4)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                    if (p >= 4 && p < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // 1 LDS read
                    if (st == 1 && p >= 8 && p < 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 LDS write
                    if (st == 0) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                // the transform
                }
                __builtin_amdgcn_sched_barrier(0);
                if (st == 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#ifdef WN_PROF
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- At M A, BN fold, activation, residuals, store: lane (q4 = l >> 4, tile l & 15) holds channels 16 ct + 4 q4 + r ----
    const int oy = 2 * ty, ox = 2 * tx;
    const bool inside = oy < H && ox < W;
    const bool row1 = oy + 1 < H;
    const int img_bytes = WN_C * HW * 4;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)n * WN_C * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.res1 ? a.res1 + (size_t)n * WN_C * HW : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.res2 ? a.res2 + (size_t)n * WN_C * HW : a.x), 0, a.res2 ? img_bytes : 0, 0x00020000);
    const unsigned lo0 = inside ? (unsigned)((4 * kq * HW + oy * W + ox) * 4) : WN_OOB;
    const unsigned lo1 = inside && row1 ? lo0 + 4u * W : WN_OOB;
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    const f32x4 sc4 = *(const f32x4*)(a.scale + 16 * ct + 4 * kq);
    const f32x4 sh4 = *(const f32x4*)(a.shift + 16 * ct + 4 * kq);
    f32x2 ra0[4], ra1[4], rb0[4], rb1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int so = (16 * ct + r) * HW * 4;
        ra0[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo0, so, 0));
        ra1[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo1, so, 0));
        rb0[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo0, so, 0));
        rb1[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo1, so, 0));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[j][r], m1 = acc[4 + j][r], m2 = acc[8 + j][r], m3 = acc[12 + j][r];
            t0[j] = m0 + m1 + m2;
            t1[j] = m1 - m2 - m3;
        }
        float o00 = t0[0] + t0[1] + t0[2], o01 = t0[1] - t0[2] - t0[3];
        float o10 = t1[0] + t1[1] + t1[2], o11 = t1[1] - t1[2] - t1[3];
        o00 = fmaf(o00, sc4[r], sh4[r]); o01 = fmaf(o01, sc4[r], sh4[r]);
        o10 = fmaf(o10, sc4[r], sh4[r]); o11 = fmaf(o11, sc4[r], sh4[r]);
        o00 = fmaxf(o00, relu_lo); o01 = fmaxf(o01, relu_lo); o10 = fmaxf(o10, relu_lo); o11 = fmaxf(o11, relu_lo);
        f32x2 q0 = {o00, o01}, q1 = {o10, o11};
        q0 += ra0[r]; q1 += ra1[r];
        q0 += rb0[r]; q1 += rb1[r];
        const int so = (16 * ct + r) * HW * 4;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q0), yr, lo0, so, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q1), yr, lo1, so, 0);
    }
#ifdef WN_PROF
    if (a.prof && lane == 0) {
        unsigned long long* d = a.prof + 4 * ((size_t)blockIdx.x * 4 + wave);
        d[0] = t_loop0 - t_entry; d[1] = t_loop1 - t_loop0; d[2] = __builtin_amdgcn_s_memtime() - t_loop1; d[3] = t_entry;
    }
#endif
}

#endif  // IC_TUNING

// all the 3x3 filters of a network in one launch (training re-packs every filter every step; 128 five-microsecond
// launches per step otherwise): layer l reads w_tab[l], writes out + l * WN_PACKED_FLOATS
__global__ __launch_bounds__(256) void wino_pack_batch_kernel(const float* const* __restrict__ w_tab, float* __restrict__ out,
                                                              int backward) {
    const float* __restrict__ w_tf = w_tab[blockIdx.y];
    float* __restrict__ o_l = out + (size_t)blockIdx.y * WN_PACKED_FLOATS;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= WN_C * WN_C) return;
    const int cin = idx / WN_C, cout = idx % WN_C;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = backward ? w_tf[(((2 - a) * 3 + (2 - b)) * WN_C + cout) * WN_C + cin]
                               : w_tf[((a * 3 + b) * WN_C + cin) * WN_C + cout];
    float t[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    float* o = o_l + (((size_t)(cout >> 5) * 64 + (cin >> 1)) * 4 * 64 + ((cin & 1) * 32 + (cout & 31))) * 4;
    float* o16 = o_l + WN_FRAG_FLOATS + (((size_t)(cout >> 4) * 32 + (cin >> 2)) * 4 * 64 + ((cin & 3) * 16 + (cout & 15))) * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        f32x4 q = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        *(f32x4*)(o + a * 256) = q;
        *(f32x4*)(o16 + a * 256) = q;
    }
}

extern "C" int ic_pack_wino3x3_c128_batch_f32(const float* const* w_tf_table_dev, float* w_packed, int layers, int backward,
                                              ic_stream_t stream) {
    IC_CHECK_ARG(w_tf_table_dev && w_packed && layers > 0);
    hipLaunchKernelGGL(wino_pack_batch_kernel, dim3(WN_C * WN_C / 256, layers), dim3(256), 0, (hipStream_t)stream,
                       w_tf_table_dev, w_packed, backward);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" size_t ic_wino3x3_c128_packed_floats(void) { return WN_PACKED_FLOATS; }

extern "C" int ic_pack_wino3x3_c128_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(WN_C * WN_C / 256), dim3(256), 0, (hipStream_t)stream, w_tf, w_packed, backward);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- launch plan -------------------------------------------------------------------------------------------------------
// No process-wide state: everything that used to be a tuning setter is a bit of the per-call `flags` word (imgcomp_hip.h,
// IC_CONV3_*).  Profiling builds (-DWN_PROF) take their stamp buffer from ic_wino3x3_c128_debug_set_prof_buffer, which does
// not exist in the product build.
#ifdef WN_PROF
static unsigned long long* g_wino_prof = nullptr;
extern "C" void ic_wino3x3_c128_debug_set_prof_buffer(void* p) { g_wino_prof = (unsigned long long*)p; }
#endif

struct WinoPlan {
    long long whole;     // tile groups run as 32 x 32 whole-K work-groups (one per CU)
    long long seg;       // tile groups run as NB-segment jobs (conv3x3_wino_tn.hip), seg_nb segments per job
    long long t16;       // tile groups run by the two-slot 16 x 16 kernel of this file
    long long ksplit;    // tile groups run K-split (4 work-groups each)
    int seg_nb;
};

// Cost model in microseconds per launch, from tools/kbench.py conv3 on the MI355X (round 2; every form, 7 shapes, interleaved):
//   whole-K round (<= 256 tile groups, one work-group per CU): 37.5 partly filled, 38.6 full, 39.8 per round in a long launch;
//   K-split round (<= 64 tile groups): 13.6, + 1.4 for the launch;
//   NB-segment jobs, one work-group per CU (NB = 3: 30.0 + 2.5 per round of 256 work-groups; NB = 2: 21 + 2.5);
//   NB = 1: two work-groups per CU, 10 per work-group a CU runs + 4.5.
// Kodak (192 groups): NB = 3 in one round 32.5 (whole-K 37.5, NB = 1 34.5, K-split 40.8); 256 groups: whole-K 38.6 (NB = 1 44);
// 272: whole-K + K-split 52.8; 4K (4050): whole-K 630 (NB = 3 716); 32 groups: NB = 1 14.3 (K-split 15.0).
static double seg_cost(long long groups, int nb) {
    if (groups <= 0) return 0.0;
    const long long wgs = 2 * ((2 * groups + nb - 1) / nb);
    if (nb == 1) return 10.0 * (double)((wgs + 255) / 256) + 4.5;
    return (double)((wgs + 255) / 256) * (nb == 3 ? 30.0 : 21.0) + 2.5;
}
static double whole_cost(long long groups) {
    if (groups <= 0) return 0.0;
    const long long rounds = (groups + 255) / 256;
    return rounds == 1 ? (groups == 256 ? 38.6 : 37.5) : 39.8 * rounds;
}
static double ksplit_cost(long long groups) { return groups <= 0 ? 0.0 : 13.6 * (double)((groups + 63) / 64) + 1.4; }

static WinoPlan wino_plan(long long groups, bool even_w, int flags) {
    WinoPlan p{0, 0, 0, 0, 0};
    const int form = flags & IC_CONV3_FORM_MASK;
    const bool leave_idle = (flags & IC_CONV3_LEAVE_IDLE_CUS) != 0;
    switch (form) {
        case IC_CONV3_WINO_WHOLEK: case IC_CONV3_WINO_WHOLEK_PW: p.whole = groups; return p;
        case IC_CONV3_WINO_KSPLIT: p.ksplit = groups; return p;
#ifdef IC_TUNING
        case IC_CONV3_WINO_T16: if (even_w) p.t16 = groups; else p.whole = groups; return p;
#endif
        case IC_CONV3_WINO_SEG1: case IC_CONV3_WINO_SEG2: case IC_CONV3_WINO_SEG3:
            if (even_w) { p.seg = groups; p.seg_nb = form - IC_CONV3_WINO_SEG1 + 1; } else p.whole = groups;
            return p;
#ifdef IC_TUNING
        case IC_CONV3_WINO_PAIR:
            if (even_w) { p.seg = groups; p.seg_nb = -1; } else p.whole = groups;      // seg_nb -1: tile-pair jobs (conv3x3_wino_tp.hip)
            return p;
#endif
        default: break;
    }
    // automatic: the full rounds of 256 tile groups run whole-K; the remainder r takes the cheapest of {whole-K, K-split,
    // NB-segment jobs as a second launch}.  Odd widths have the per-wave whole-K kernel and K-split only; next to a caller's
    // CU-range stream (leave_idle) only the one-work-group-per-CU forms whose work-groups stay below 256 qualify.
    // the caller keeps n independent launches of this shape in flight: when they cover the chip together, every one of them runs
    // the form with the lowest CU-time per tile group (whole-K: one work-group per group for 37.5 us; NB = 3 segments: 1.33 CUs
    // per group for 30 us + its boundary) -- measured on Kodak-sized images, 4 in flight: 183 against 168 Mpix/s (bench.py)
    const int in_flight = (flags >> 19) & 0xf;
    // (launches of >= 128 groups only: on a 64 x 64 map -- 32 work-groups of 36 us each -- the layer chain of one image gets too
    // long to be covered by the other images: 66 against 108 Mpix/s at 8 in flight)
    if (in_flight >= 2 && groups >= 128 && !leave_idle) { p.whole = groups; return p; }
    const long long r = groups % 256, full = groups - r;
    double best = whole_cost(r);
    p.whole = groups;
    if (r > 0 && ksplit_cost(r) < best && (!leave_idle || full > 0 || 4 * r <= 256)) {
        best = ksplit_cost(r); p = WinoPlan{full, 0, 0, r, 0};
    }
    if (r > 0 && even_w && !leave_idle) {
        for (int nb = 1; nb <= 3; ++nb) {
            const double c = seg_cost(r, nb) + (full > 0 ? 1.0 : 0.0);
            if (c < best) { best = c; p = WinoPlan{full, r, 0, 0, nb}; }
        }
    }
    return p;
}

// work-groups-per-CU footprint of the launch(es) ic_wino3x3_c128_bn_act_f32 would make for this shape: the number of CUs a
// caller sizing a CU-range stream for an independent branch has to leave to this layer (256 = the whole chip)
extern "C" long long ic_wino3x3_c128_workgroups(int N, int H, int W, int flags) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const WinoPlan p = wino_plan((long long)N * ic_cdiv(H, 4) * ic_cdiv(W, 32), (W & 1) == 0, flags);
    long long cus = p.whole + 4 * p.ksplit;                          // one 512-register work-group per CU
    if (p.seg > 0 && p.seg_nb < 0) cus += 2 * p.seg;                 // tile-pair jobs: 4 work-groups per group, two per CU
    else if (p.seg > 0) {
        const long long wgs = 2 * ((2 * p.seg + p.seg_nb - 1) / p.seg_nb);
        cus += p.seg_nb > 1 ? wgs : (wgs + 1) / 2;                   // NB = 1: two work-groups share a CU
    }
    cus += (4 * p.t16 + 1) / 2;
    return cus;
}

// 1: this build carries the forms that the plan never picks by itself (16 x 16 jobs, tile pairs, the persistent stack kernel)
extern "C" int ic_build_has_tuning_forms(void) {
#ifdef IC_TUNING
    return 1;
#else
    return 0;
#endif
}

// the plan itself: tile groups per form {whole-K, NB-segment, segments per job NB, 16 x 16 (two-slot), K-split}
extern "C" int ic_wino3x3_c128_plan(int N, int H, int W, int flags, long long plan_out[5]) {
    IC_CHECK_ARG(plan_out && N > 0 && H > 0 && W > 0);
    const WinoPlan p = wino_plan((long long)N * ic_cdiv(H, 4) * ic_cdiv(W, 32), (W & 1) == 0, flags);
    plan_out[0] = p.whole; plan_out[1] = p.seg; plan_out[2] = p.seg_nb; plan_out[3] = p.t16; plan_out[4] = p.ksplit;
    return IC_OK;
}

extern "C" int ic_wino3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                          const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                                          int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y);
    IC_CHECK_ARG(N > 0 && H > 0 && W > 0);
    if ((long long)WN_C * H * W * 4 >= (1ll << 31)) return IC_ERR_UNSUPPORTED;    // per-image byte offsets are 31-bit
#ifndef IC_TUNING
    {   // the 16 x 16-job and tile-pair forms are compiled into tuning builds only (ic_build_has_tuning_forms)
        const int form = flags & IC_CONV3_FORM_MASK;
        if (form == IC_CONV3_WINO_T16 || form == IC_CONV3_WINO_PAIR) return IC_ERR_UNSUPPORTED;
    }
#endif
    WnArgs a{};
    a.x = x; a.wp = w_packed; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    a.grows = ic_cdiv(H, 4); a.gcols = ic_cdiv(W, 32); a.xcd_runs = (flags & IC_CONV3_NO_XCD_RUNS) ? 0 : 1;
#ifdef WN_PROF
    a.prof = g_wino_prof;
#endif
    const bool even_w = (W & 1) == 0;
    const WinoPlan p = wino_plan((long long)N * a.grows * a.gcols, even_w, flags);
    hipStream_t st = (hipStream_t)stream;
    long long g0 = 0;
    if (p.whole > 0) {
        const dim3 grid((unsigned)p.whole);
        a.g0 = (int)g0; a.ngroups = (int)p.whole;
        if (even_w && (flags & IC_CONV3_FORM_MASK) != IC_CONV3_WINO_WHOLEK_PW) {
            if (p.whole <= 256) hipLaunchKernelGGL(wino3x3_c128_shared_kernel<true>, grid, dim3(256), 0, st, a);      // one round
            else hipLaunchKernelGGL(wino3x3_c128_shared_kernel<false>, grid, dim3(256), 0, st, a);
        }
        else if (even_w) hipLaunchKernelGGL(wino3x3_c128_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(wino3x3_c128_kernel<false>, grid, dim3(256), 0, st, a);
        g0 += p.whole;
    }
    if (p.seg > 0) {
        a.g0 = (int)g0; a.ngroups = (int)p.seg;
#ifdef IC_TUNING
        const int rc = p.seg_nb < 0 ? icx_wino_tp_launch(a, st) : icx_wino_tn_launch(a, p.seg_nb, (flags & IC_CONV3_PACKED_TRANSFORM) ? 0 : 1, st);
#else
        const int rc = icx_wino_tn_launch(a, p.seg_nb, (flags & IC_CONV3_PACKED_TRANSFORM) ? 0 : 1, st);
#endif
        if (rc) return rc;
        g0 += p.seg;
    }
#ifdef IC_TUNING
    if (p.t16 > 0) {
        a.g0 = (int)g0; a.ngroups = (int)p.t16;
        hipLaunchKernelGGL(wino3x3_c128_t16_kernel, dim3((unsigned)(4 * p.t16)), dim3(256), 0, st, a);
        g0 += p.t16;
    }
#endif
    if (p.ksplit > 0) {
        const dim3 grid((unsigned)(p.ksplit * 4));
        a.g0 = (int)g0; a.ngroups = (int)p.ksplit;
        if (even_w) hipLaunchKernelGGL(wino3x3_c128_ksplit_kernel<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(wino3x3_c128_ksplit_kernel<false>, grid, dim3(256), 0, st, a);
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- both forms behind one packed filter: what the network / training entry points use ---------------------------------
// blob = [direct-form fragments (ic_conv3x3_c128_packed_floats) | Winograd fragments (2 x 16 x 128 x 128)].
// The form is picked per launch: Winograd whenever the shape is addressable with 31-bit offsets, the direct kernel
// otherwise or on request (flags & IC_CONV3_FORM_MASK == IC_CONV3_DIRECT).
// blob = [direct-form fragments | F(2x2) fragments (2 x 16 x 128 x 128) | F(4x4) fragments (36 x 128 x 128)]
extern "C" size_t ic_conv3x3_c128_both_packed_floats(void) {
    return ic_conv3x3_c128_packed_floats() + WN_PACKED_FLOATS + ic_wino4_3x3_c128_packed_floats();
}

extern "C" int ic_pack_conv3x3_c128_both_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    int rc = backward ? ic_pack_conv3x3_c128_bwd_f32(w_tf, w_packed, stream) : ic_pack_conv3x3_c128_f32(w_tf, w_packed, stream);
    if (rc) return rc;
    if ((rc = ic_pack_wino3x3_c128_f32(w_tf, w_packed + ic_conv3x3_c128_packed_floats(), backward, stream))) return rc;
    return ic_pack_wino4_3x3_c128_f32(w_tf, w_packed + ic_conv3x3_c128_packed_floats() + WN_PACKED_FLOATS, backward, stream);
}

extern "C" int ic_conv3x3_c128_pick_algo(int N, int H, int W, int flags) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    if ((long long)WN_C * H * W * 4 >= (1ll << 31)) return 0;
    // Measured on the MI355X: the direct kernel has a ~27 us floor (16 sequential channel chunks per work-group) and
    // 95-105 TFLOP/s at best; the Winograd forms win at every shape they can address.
    return (flags & IC_CONV3_FORM_MASK) == IC_CONV3_DIRECT ? 0 : 1;
}

// The form ic_conv3x3_c128_auto_f32 runs: 0 direct, 1 Winograd F(2x2,3x3) (decomposition: ic_wino3x3_c128_plan), 2 Winograd F(4x4,3x3).
// F(4x4) executes 0.5625 of F(2x2)'s multiplies and streams 2.25 x its filter fragments (2.36 MB per layer, cold in L2 at every layer
// of a network: 64 layers x 2.36 MB).  It is chosen
//   * for a launch of >= 160 work-groups (a Kodak map: 192), alone or not -- round 5, after the filter ring went from 6 to 9 quads
//     and a lone launch began to pull the NEXT layer's fragments into L2 (conv3x3_wino4.hip).  One image at a time through the whole
//     network (bench.py --in_flight 1, automatic F(2x2) plan against F(4x4) forced, Mpix/s):
//         384x512 (96 work-groups) 111.5 / 91.1    512x512 (128) 143.6 / 119.4    448x768 (168) 145.3 / 150.2    512x768 (192) 162.8 / 169.3
//         640x768 (240) 174.0 / 201.9    768x768 (288) 152.7 / 162.4    768x1024 (384) 170.6 / 213.0
//     (with the patch requested 10 quads ahead of its transform, later in round 5: 512x768 175.6, 640x768 210.4, 768x1024 216.3)
//     (round 4, 6-quad ring, no prefetch: 512x768 162.5 / 140.7 -- the threshold stood at 512 work-groups, two per CU, then);
//   * with several independent launches in flight (IC_CONV3_IN_FLIGHT: the images of an evaluation set) from 384 work-groups together;
//   * and only where the map fills its 16-tile segments (1 x 16 tiles, or 2 x 8 on narrow maps: whichever needs fewer): 30 maps of
//     40 x 40 (62 % full) 68 against 95 us; below 55 % the F(2x2) plan stays.
extern "C" int ic_conv3x3_c128_pick_form(int N, int H, int W, int flags) {
    if (ic_conv3x3_c128_pick_algo(N, H, W, flags) == 0) return 0;
    const int form = flags & IC_CONV3_FORM_MASK;
    if (form == IC_CONV3_WINO4) return ic_wino4_3x3_c128_supported(N, H, W) ? 2 : 1;
    if (form != IC_CONV3_AUTO && form != IC_CONV3_WINO) return 1;                  // a particular F(2x2) decomposition was asked for
    if (!ic_wino4_3x3_c128_supported(N, H, W) || (flags & (IC_CONV3_LEAVE_IDLE_CUS | IC_CONV3_NO_WINO4))) return 1;
    const long long wgs = ic_wino4_3x3_c128_workgroups(N, H, W);
    const long long tiles = (long long)N * ic_cdiv(H, 4) * ic_cdiv(W, 4);
    if (tiles * 100 < wgs * 8 * 55) return 1;                                     // segments less than 55 % full (16 tiles x 2 halves per segment)
    const int in_flight = (flags >> 19) & 0xf;
    return (wgs >= 160 || (in_flight >= 2 && wgs * in_flight >= 384)) ? 2 : 1;
}

int icx_conv3x3_c128_auto_next(const float* x, const float* w_both, const float* scale, const float* shift,
                               const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                               int flags, const float* w_both_next, hipStream_t stream) {
    IC_CHECK_ARG(x && w_both && scale && shift && y && N > 0 && H > 0 && W > 0);
    const int form = ic_conv3x3_c128_pick_form(N, H, W, flags);
    if (form == 2) {
        const size_t f4 = ic_conv3x3_c128_packed_floats() + WN_PACKED_FLOATS;      // the F(4x4) fragments inside a `both` blob
        // prefetch of the next layer's fragments: one call at a time only (conv3x3_wino4.hip has the measurements)
        const bool alone = ((flags >> 19) & 0xf) < 2;
        return icx_wino4_3x3_c128_next(x, w_both + f4, scale, shift, res1, res2, y, N, H, W, relu, flags,
                                       (w_both_next && alone) ? w_both_next + f4 : nullptr, stream);
    }
    if (form == 1)
        return ic_wino3x3_c128_bn_act_f32(x, w_both + ic_conv3x3_c128_packed_floats(), scale, shift, res1, res2, y, N, H, W,
                                          relu, flags & ~IC_CONV3_WINO4_BITS, stream);
    return ic_conv3x3_c128_bn_act_f32(x, w_both, scale, shift, res1, res2, y, N, H, W, relu, flags, stream);
}
extern "C" int ic_conv3x3_c128_auto_f32(const float* x, const float* w_both, const float* scale, const float* shift,
                                        const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                                        int flags, ic_stream_t stream) {
    return icx_conv3x3_c128_auto_next(x, w_both, scale, shift, res1, res2, y, N, H, W, relu, flags, nullptr, (hipStream_t)stream);
}
