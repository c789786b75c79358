// 3x3, stride 1, 128 -> 128 channel convolution of the residual stacks (autoencoder.py:274-287), Winograd F(2x2,3x3), for
// callers that keep several independent images in flight (IC_CONV3_IN_FLIGHT): "tile-pair, position-split" jobs, TWO
// work-groups per CU.
//
// Why.  With images in flight a launch need not fill the chip, so the form with the least CU-time wins -- so far the 32 x 32
// whole-K form (conv3x3_wino.hip): 93 % matrix-pipe issue inside its k-loop, but 256 accumulator registers per wave = ONE
// work-group per CU, so its prologue, epilogue and the turnover to the next work-group (2.9 k + 8.4 k + ~6 k of 88 k clocks)
// run with the matrix pipes idle: 0.745 of the peak of the CUs it occupies.  Two co-resident work-groups in different phases
// hide each other's head and tail -- that needs <= 256 registers per wave and <= 80 KB of LDS per work-group.  The NB = 1
// segment form (conv3x3_wino_tn.hip) has that footprint but streams every filter fragment for 16 tiles only (393 MB of
// L2 -> L1 reads per Kodak launch: its bound).  Here:
//
//   a work-group = one TILE GROUP (2 segments = 2 x 16 tiles: the two tile rows of a group) x 32 output channels;
//   wave w = channel tile (w & 1) x POSITION HALF (w >> 1): 16 channels x 32 tiles x 8 of the 16 Winograd positions
//       = 2 x 8 accumulators of 4 registers = 64 AGPRs; 32 k-steps (4 input channels) of 16 v_mfma_f32_16x16x4_f32;
//   every filter fragment a wave loads multiplies 32 tiles (the whole-K form's reuse: 268 MB of L2 -> L1 per Kodak launch);
//   the input transform is shared by the four waves (wave w transforms k-steps 4 j + w of both segments) through a TWO-slot
//   LDS ring (64 KB): one barrier per 4 k-steps, placed before the last block of an iteration -- every read of the current
//   slot and every write of the next one precede it, the first operands of the next iteration are read behind it;
//   At M A needs all 16 positions of a (channel, tile): the waves of position half 1 hand their raw sums to their partner
//   through LDS once, at the end, and the partner runs the same epilogue as every other Winograd form -- the same
//   operations per output in the same order, so results are bit-identical to them (tested).
//
// MEASURED (round 3), and why this form is NOT what the plan picks: Kodak layer 37.0 us alone (whole-K 36.0, NB = 3 32.1), 160
// against 186 Mpix/s with six images in flight.  Ablation builds (-DTP_ABL): without the end-of-row / all patch re-requests
// 36.6 / 36.5 us, without the filter requests 36.7, without the input transform **31.2**, with nothing but MFMAs and the
// barrier 30.5 (= 2 rounds of 13.7 us: the matrix pipes then run at ~100 %).  The two resident work-groups do hide each other's
// memory latency, head and tail -- but a vector instruction takes the SIMD's issue slot from the OTHER wave's MFMA just as
// it does from its own (80 transform instructions per 64 MFMAs cost 21 % here, 4 x the whole-K form's ratio, because only
// 32 output channels share a transform).  Two work-groups per CU only pay with the whole-K form's vector work per MFMA clock,
// and that needs all 128 output channels behind one transform: 128 ch x 16 tiles x 16 positions = 128 accumulator registers
// per wave is the 16 x 16-job form again, with twice the filter stream.  Kept as a selectable, tested form (IC_CONV3_WINO_PAIR).
#include "wino_common.h"
#include "internal.h"

#define TP_FST 4                  // filter ring: requested 3 k-steps ahead
#ifndef TP_ABL
#define TP_ABL 0                  // tuning builds (wrong results): 1 no end-of-row re-requests, 2 no patch re-requests at all, 4 no transform, 8 no filter requests, 16 no B reads
#endif

__device__ __forceinline__ float tp_add(float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float tp_sub(float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }

template <bool WT>
__global__ __launch_bounds__(256, 2) void wino3x3_c128_tp_kernel(const WnArgs a) {
    constexpr int NB = 2;
    constexpr int SLOT = 4 * NB * 4 * 64;                       // f32x4 elements of one ring slot: [k-step][segment][quad][lane]
    __shared__ f32x4 ring[2 * SLOT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kq = lane >> 4, tj = lane & 15;
    const int ctl = wave & 1, ph = wave >> 1;
    // work-group b: tile group (b >> 2) x channel quarter (b & 3); the four quarters of a group sit next to each other in the
    // XCD's contiguous run: they read the same input
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int ct = 2 * (b & 3) + ctl;                           // 16-channel output tile of this wave, 0 .. 7
    const int H = a.H, W = a.W, HW = H * W;
    int n, gy, gx;
    {
        const int g = a.g0 + (b >> 2);
        const int t = a.mg_cols ? (int)__umulhi((unsigned)g, a.mg_cols) : g;            // g / gcols
        gx = g - t * a.gcols;
        n = a.mg_rows ? (int)__umulhi((unsigned)t, a.mg_rows) : t;                      // t / grows
        gy = t - n * a.grows;
    }
    const int tx = gx * 16 + tj;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * WN_C * HW), 0, WN_C * HW * 4, 0x00020000);
    f32x2 pp[NB][4], pe[NB][4];                                  // own pair / end-of-row pair of the 4 patch rows
    unsigned o0[NB][4], oe[NB][4];
    {
        const int ecol = 2 * tx + (tj == 0 ? -2 : 2);
        const bool has_e = (tj == 0 || tj == 15) && ecol >= 0 && ecol < W;
        const bool has_0 = 2 * tx < W;
        const int so = wave * 4 * HW * 4;                       // own k-step of the first ring slot
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r0 = 2 * (2 * gy + i) - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const bool rok = r >= 0 && r < H;
                const unsigned rb = (unsigned)(kq * HW + r * W) * 4u;
                o0[i][q] = (rok && has_0) ? rb + 8u * tx : WN_OOB;
                oe[i][q] = (rok && has_e) ? rb + 4u * ecol : WN_OOB;
                pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, o0[i][q], so, 0));
                pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, oe[i][q], so, 0));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // filter fragments of this wave: quads 2 ph, 2 ph + 1 (positions 8 ph .. 8 ph + 7) of channel tile ct, 16-channel-tile packing
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
    const unsigned fo = lane * 16u + (unsigned)ph * 2048u;
    f32x4 fl[TP_FST][2];
#pragma unroll
    for (int st = 0; st < TP_FST - 1; ++st) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            fl[st][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + q * 1024u, (ct * 32 + st) * 4096, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NB][8];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][p][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 8; ++p) asm volatile("" : "+a"(acc[i][p]));
    __builtin_amdgcn_sched_barrier(0);

    // Bt d B of one lane's patch -> the 16 B operands of a k-step (position 4 row + column), in 16 micro-steps (as the
    // NB-segment kernel: outer columns by DPP, rows, columns)
    float tc[4][4], tu[4][4];
    auto tr_step = [&](int sg, int s, float (&v)[16]) __attribute__((always_inline)) {
        if (s < 4) {
            const int q = s;
            tc[1][q] = pp[sg][q][0]; tc[2][q] = pp[sg][q][1];
            tc[3][q] = dpp_from_right(pe[sg][q][0], tc[1][q]);
            tc[0][q] = dpp_from_left(pe[sg][q][1], tc[2][q]);
        } else if (s < 12) {
            const int k = (s - 4) >> 1;
            if (((s - 4) & 1) == 0) { tu[k][0] = tp_sub(tc[k][0], tc[k][2]); tu[k][1] = tp_add(tc[k][1], tc[k][2]); }
            else { tu[k][2] = tp_sub(tc[k][2], tc[k][1]); tu[k][3] = tp_sub(tc[k][1], tc[k][3]); }
        } else {
            const int q = s - 12;
            v[4 * q] = tp_sub(tu[0][q], tu[2][q]); v[4 * q + 1] = tp_add(tu[1][q], tu[2][q]);
            v[4 * q + 2] = tp_sub(tu[2][q], tu[1][q]); v[4 * q + 3] = tp_sub(tu[1][q], tu[3][q]);
        }
    };
    f32x4* const rb0 = ring + lane;
    constexpr int NKS = 32, NIT = NKS / 4;
    float vt[16];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int s = 0; s < 16; ++s) tr_step(i, s, vt);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 tq = {vt[4 * q], vt[4 * q + 1], vt[4 * q + 2], vt[4 * q + 3]};
            rb0[((wave * NB + i) * 4 + q) * 64] = tq;
        }
        const int so = (4 + wave) * 4 * HW * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, o0[i][q], so, 0));
            pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, oe[i][q], so, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    f32x4 bq[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) bq[0][q] = rb0[(2 * ph + q) * 64];

    // Issue order as in the NB-segment kernel: inline-asm MFMA with the accumulator tied ("+a"), one memory / LDS instruction
    // and a few vector instructions behind it, pinned by sched_barrier.  A block = (k-step st, segment i) = 8 MFMAs.
    //   block (st, 0), MFMAs 0-1: the k-step's two filter requests          block (*), MFMAs 2-3: B operands of the next block
    //   blocks 0-1: transform of segment 0 (own k-step of iteration j + 1), blocks 2-3: segment 1
    //   block 2 / 4, MFMAs 4-7: ring writes of segment 0 / 1;  blocks 3 / 5: their patch registers re-requested (iteration j + 2)
    f32x4* rd = rb0; f32x4* wr = rb0 + SLOT;
    for (int j = 0; j < NIT; ++j) {
        const int kp = 4 * (j + 2) + wave < NKS ? 4 * (j + 2) + wave : NKS - 4 + wave;     // own k-step two iterations on
        const int kf = 4 * j + TP_FST - 1;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                constexpr int LAST = 4 * NB - 1;
                const int blk = st * NB + i;
                const int fs = (st + TP_FST - 1) % TP_FST;
                const int fso = (ct * 32 + (kf + st < NKS ? kf + st : NKS - 1)) * 4096;
                if (blk == LAST) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4* nsl = blk < LAST ? rd + (((blk + 1) * 4 + 2 * ph) * 64) : wr + (2 * ph) * 64;     // next block's quads
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][p]) : "v"(fl[st][p >> 2][p & 3]), "v"(bq[blk & 1][p >> 2][p & 3]));
                    if (i == 0 && p < 2 && !(TP_ABL & 8))
                        fl[fs][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + p * 1024u, fso, 0));
                    if (p >= 2 && p < 4 && !(TP_ABL & 16)) bq[(blk + 1) & 1][p - 2] = nsl[(p - 2) * 64];
                    if (blk < 4 && !(TP_ABL & 4)) tr_step(blk >> 1, (blk & 1) * 8 + p, vt);  // transform of segment blk / 2: one micro-step per MFMA
                    if ((blk == 2 || blk == 4) && p >= 4) {                  // ring writes of segment (blk - 2) / 2
                        const int sg = (blk - 2) >> 1, q = p - 4;
                        const f32x4 tq = {vt[4 * q], vt[4 * q + 1], vt[4 * q + 2], vt[4 * q + 3]};
                        wr[((wave * NB + sg) * 4 + q) * 64] = tq;
                    }
                    if (blk == 3 || blk == 5) {                              // patch re-request of segment (blk - 3) / 2
                        const int sg = (blk - 3) >> 1, q = p >> 1, so = kp * 4 * HW * 4;
                        if ((p & 1) == 0) { if (!(TP_ABL & 2)) pp[sg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, o0[sg][q], so, 0)); }
                        else if (!(TP_ABL & 3)) pe[sg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, oe[sg][q], so, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        f32x4* const t = rd; rd = wr; wr = t;
    }
    // ---- hand-over of position half 1, then the common epilogue on position half 0 ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 8; ++p) asm volatile("" : "+a"(acc[i][p]));
    __syncthreads();                                             // nobody reads the ring any more
    f32x4* const xch = ring + (ctl * 16) * 64 + lane;            // [channel tile][segment][position 0..7][lane]
    if (ph == 1) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int p = 0; p < 8; ++p) xch[(i * 8 + p) * 64] = acc[i][p];
    }
    __syncthreads();
    if (ph == 1) return;
    f32x4 hi[NB][8];                                             // positions 8 .. 15 from the partner wave
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 8; ++p) hi[i][p] = xch[(i * 8 + p) * 64];

    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    const f32x4 sc4 = *(const f32x4*)(a.scale + 16 * ct + 4 * kq);
    const f32x4 sh4 = *(const f32x4*)(a.shift + 16 * ct + 4 * kq);
    const int img_bytes = WN_C * HW * 4;
    const size_t ib = (size_t)n * WN_C * HW;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + ib), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res1 ? a.res1 + ib : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res2 ? a.res2 + ib : a.x), 0, a.res2 ? img_bytes : 0, 0x00020000);
    unsigned lo0[NB], lo1[NB];
    f32x2 ra0[NB][4], ra1[NB][4], rb0v[NB][4], rb1v[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int oy = 2 * (2 * gy + i), ox = 2 * tx;
        const bool inside = oy < H && ox < W;
        lo0[i] = inside ? (unsigned)((4 * kq * HW + oy * W + ox) * 4) : WN_OOB;
        lo1[i] = inside && oy + 1 < H ? lo0[i] + 4u * W : WN_OOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int so = (16 * ct + r) * HW * 4;
            ra0[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo0[i], so, 0));
            ra1[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo1[i], so, 0));
            rb0v[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo0[i], so, 0));
            rb1v[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo1[i], so, 0));
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0[4], t1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[i][c][r], m1 = acc[i][4 + c][r], m2 = hi[i][c][r], m3 = hi[i][4 + c][r];
                t0[c] = m0 + m1 + m2;
                t1[c] = m1 - m2 - m3;
            }
            float o00 = t0[0] + t0[1] + t0[2], o01 = t0[1] - t0[2] - t0[3];
            float o10 = t1[0] + t1[1] + t1[2], o11 = t1[1] - t1[2] - t1[3];
            o00 = fmaf(o00, sc4[r], sh4[r]); o01 = fmaf(o01, sc4[r], sh4[r]);
            o10 = fmaf(o10, sc4[r], sh4[r]); o11 = fmaf(o11, sc4[r], sh4[r]);
            o00 = fmaxf(o00, relu_lo); o01 = fmaxf(o01, relu_lo); o10 = fmaxf(o10, relu_lo); o11 = fmaxf(o11, relu_lo);
            f32x2 v0 = f32x2{o00, o01} + ra0[i][r], v1 = f32x2{o10, o11} + ra1[i][r];
            v0 += rb0v[i][r]; v1 += rb1v[i][r];
            const int so = (16 * ct + r) * HW * 4;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v0), yr, lo0[i], so, WT ? 16 : 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v1), yr, lo1[i], so, WT ? 16 : 0);
        }
    }
}

int icx_wino_tp_launch(const WnArgs& a_in, hipStream_t st) {
    if (a_in.ngroups <= 0) return IC_OK;
    if (a_in.W & 1) return IC_ERR_UNSUPPORTED;
    WnArgs a = a_in;
    if ((unsigned long long)(a.g0 + a.ngroups) * (unsigned)(a.gcols > a.grows ? a.gcols : a.grows) >= (1ull << 32)) return IC_ERR_UNSUPPORTED;
    a.mg_cols = a.gcols > 1 ? (unsigned)((1ull << 32) / (unsigned)a.gcols) + 1u : 0u;
    a.mg_rows = a.grows > 1 ? (unsigned)((1ull << 32) / (unsigned)a.grows) + 1u : 0u;
    const dim3 grid((unsigned)(4 * a.ngroups)), block(256);
    if (4 * a.ngroups <= 512) hipLaunchKernelGGL(wino3x3_c128_tp_kernel<true>, grid, block, 0, st, a);       // one resident round
    else hipLaunchKernelGGL(wino3x3_c128_tp_kernel<false>, grid, block, 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
