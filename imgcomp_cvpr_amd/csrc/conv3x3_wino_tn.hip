// 3x3, stride 1, 128 -> 128 channel convolution of the residual stacks (autoencoder.py:274-287) in Winograd F(2x2,3x3) form,
// "NB-segment" jobs on v_mfma_f32_16x16x4_f32:
//
//   a SEGMENT is 16 horizontally adjacent 2x2-pixel tiles of one tile row (2 x 32 output pixels);
//   a wave-job is 16 output channels x NB segments x 16 transform positions x all 128 input channels:
//       32 k-steps (4 input channels each) of 16 NB MFMAs, NB x 64 accumulator registers, no cross-wave reduction;
//   a work-group is 4 waves = 4 channel tiles (one HALF of the output channels) of the same NB segments.
//
// Why this shape.  The 32 x 32 whole-K form (conv3x3_wino.hip) makes wave-jobs of 65.5 k matrix-pipe clocks and one
// work-group per CU: a Kodak map is 768 of them on 1024 SIMDs, a quarter of the chip idles for the whole launch.  16 x 16 jobs
// (NB = 1) balance -- 3072 jobs, three per SIMD -- but every wave streams its own 128 KB of filter fragments for only 8192
// short MFMAs: 393 MB of L2 -> L1 reads per Kodak launch, which bounds that form.  Here one filter fragment (A operand) is
// multiplied with the B operands of NB segments before the next one is needed: NB = 3 makes a Kodak map (384 segments) exactly
// 128 x 2 = 256 work-groups = 1024 wave-jobs of 49.2 k matrix-pipe clocks, ONE per SIMD, and a third of the filter stream per
// MFMA.  NB = 2 / NB = 1 serve the shapes whose segment count balances better that way (wino_plan in conv3x3_wino.hip).
//
// Operand flow per k-step, as in the other Winograd kernels: the B operands are the input transform Bt d B of each lane's
// tile -- lane l holds input channel 4 ks + (l >> 4) of tile (l & 15): four aligned pixel pairs (raw buffer loads, zero
// padding by out-of-range offsets), the outer patch columns from the neighbour lanes by DPP row shifts, 32 add/subs.
// The four waves of the work-group multiply the SAME B operands with different channel tiles, so wave w loads and transforms
// only k-steps 4 j + w and the operands reach the other waves through an LDS ring.  The ring has THREE slots of four k-steps:
// iteration j reads slot j % 3 and writes slot (j + 1) % 3, one barrier per iteration between the last write of the
// iteration and the first read of the next slot; a slot is rewritten two barriers after its last read, so B operands are
// read one 16-MFMA block ahead of their use all the way through the iteration (a two-slot ring would need all of the
// iteration's last k-step -- 16 NB registers -- in flight before the barrier).
// The A operands stream from L2 in the 16-channel-tile fragment order of the packed filter (ic_pack_wino3x3_c128_f32),
// three k-steps ahead.  Outputs are bit-identical to the other Winograd forms: the same operations per output.
#include "wino_common.h"
#include "internal.h"

#define TN_FST 4                  // filter ring: requested 3 k-steps ahead
#ifndef TN_TR_AT
#define TN_TR_AT 12               // the 40 vector instructions of a segment's input transform go behind MFMAs TN_TR_AT .. +TN_TR_N-1
#define TN_TR_N 4                 // of its block as bursts (measured: 1 us per Kodak layer faster than 2-4 behind every MFMA)
#endif
#ifndef TN_SCHED
#define TN_SCHED 0                // main-loop issue order: 0 = fillers spread over the MFMA gaps, 1 = clustered into bursts
#endif
#ifndef TN_P_FILT
#define TN_P_FILT 0               // clustered schedule: the gap (MFMA index of the block) each burst goes behind
#define TN_P_READ 4
#define TN_P_TR 8
#define TN_P_PUT 12
#define TN_TR_GAPS 1              // transform burst in 1, 2 or 4 gaps
#endif
#ifndef TN_ABL
#define TN_ABL 0                  // tuning builds: 1 no transform, 2 no patch re-requests, 4 no filter requests, 8 no B reads, 16 no ring writes, 32 no barrier
#endif

#ifdef TN_INT_OPS       // tuning builds: integer adds instead of fp32 adds (wrong results; is the cost the FP32 ALU or the issue slot?)
__device__ __forceinline__ float tn_add(float x, float y) { float r; asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float tn_sub(float x, float y) { float r; asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
#else
__device__ __forceinline__ float tn_add(float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float tn_sub(float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
#endif

// PK: input transform on packed-fp32 adds (v_pk_add_f32: 16 instructions instead of 32); PK = false keeps every add a
// single-issue v_add_f32 / v_sub_f32 (inline asm, so that the SLP vectoriser does not re-pack them) -- packed fp32 next to
// MFMAs is measured as slower per instruction than two plain adds (MI355X_MICROARCH.md, "price of one filler").
template <int NB, bool PK, bool WT>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void wino3x3_c128_tn_kernel(const WnArgs a) {
#ifdef WN_PROF
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
#ifdef WN_PROF3     // shader clock against the constant 100 MHz counter: what frequency does the kernel actually run at?
    const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int SLOT = 4 * NB * 4 * 64;                       // float4 elements of one ring slot: [k-step][segment][quad][lane]
    __shared__ f32x4 ring[3 * SLOT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kq = lane >> 4, tj = lane & 15;
    // work-group b: segments [NB job, NB job + NB) of this launch x channel half hc.  Both halves of a job sit next to each
    // other in the XCD's contiguous run: they read the same input.
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int hc = b & 1, job = b >> 1;
    const int ct = 4 * hc + wave;                               // 16-channel output tile of this wave
    const int H = a.H, W = a.W, HW = H * W;
    const int nseg = 2 * a.ngroups;

    // ---- the NB segments of this job: segment s of the launch = tile row (s & 1) of tile group g0 + s / 2 ----
    // Everything up to the first patch request is serial scalar work of a wave that has nothing else to run (measured:
    // 2,700 clocks with compiler-generated integer divisions, branches and all address arithmetic ahead of the first load;
    // the layer is 74,000).  So: divisions by host-made reciprocals, no branches, and each segment's requests leave as soon
    // as ITS addresses exist.
    int sg_n[NB], sg_ty[NB], sg_gx[NB];
    bool sg_ok[NB];
    {
        const int s0 = job * NB;
        const int g = a.g0 + (s0 >> 1);
        const int t = a.mg_cols ? (int)__umulhi((unsigned)g, a.mg_cols) : g;            // g / gcols
        int gx = g - t * a.gcols;
        int n = a.mg_rows ? (int)__umulhi((unsigned)t, a.mg_rows) : t;                  // t / grows
        int gy = t - n * a.grows, r = s0 & 1;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            sg_ok[i] = s0 + i < nseg;
            sg_n[i] = sg_ok[i] ? n : 0; sg_ty[i] = 2 * gy + r; sg_gx[i] = gx;
            r ^= 1;
            gx += r == 0;                                        // next tile group after its second row
            const bool wx = gx == a.gcols; gx = wx ? 0 : gx;
            gy += wx;
            const bool wy = gy == a.grows; gy = wy ? 0 : gy;
            n += wy;
        }
    }
    f32x2 pp[NB][4], pe[NB][4];                                  // own pair / end-of-row pair of the 4 patch rows
    __amdgpu_buffer_rsrc_t xr[NB];
    unsigned o0[NB][4], oe[NB][4];
#ifdef WN_PROF2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t_addr = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        xr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)sg_n[i] * WN_C * HW), 0, WN_C * HW * 4, 0x00020000);
        const int tx = sg_gx[i] * 16 + tj;
        const int r0 = 2 * sg_ty[i] - 1;
        // the end-of-row value is fetched as the aligned pair that contains it -- lane 0: (2tx-2, 2tx-1), lane 15: (2tx+2, 2tx+3)
        const int ecol = 2 * tx + (tj == 0 ? -2 : 2);
        const bool has_e = (tj == 0 || tj == 15) && ecol >= 0 && ecol < W;
        const bool has_0 = 2 * tx < W;
        const int so = wave * 4 * HW * 4;                       // own k-step of the first ring slot
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            const bool rok = sg_ok[i] && r >= 0 && r < H;
            const unsigned rb = (unsigned)(kq * HW + r * W) * 4u;
            o0[i][q] = (rok && has_0) ? rb + 8u * tx : WN_OOB;
            oe[i][q] = (rok && has_e) ? rb + 4u * ecol : WN_OOB;
            pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], o0[i][q], so, 0));
            pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], oe[i][q], so, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
    const unsigned fo = lane * 16u;
    f32x4 fl[TN_FST][4];
#pragma unroll
    for (int st = 0; st < TN_FST - 1; ++st) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            fl[st][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + q * 1024u, (ct * 32 + st) * 4096, 0));
        __builtin_amdgcn_sched_barrier(0);
    }
    // the accumulators are cleared while the requests are in flight (192 v_accvgpr_write for NB = 3)
    f32x4 acc[NB][16];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][p][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[i][p]));
    __builtin_amdgcn_sched_barrier(0);

    auto load_patch = [&](int i, int ks) __attribute__((always_inline)) {
        const int so = ks * 4 * HW * 4;                         // scalar: channels 4 ks ..
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], o0[i][q], so, 0));
            pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], oe[i][q], so, 0));
        }
    };
    // Bt d B of one lane's patch -> the 16 B operands of a k-step (position 4 row + column), cut into 16 micro-steps of
    // 2-4 vector instructions so that the main loop can place them behind individual MFMAs.
    //   steps 0-3   (patch row q):     the outer columns from the neighbour lanes (DPP row shifts; row ends keep the edge pair)
    //   steps 4-11  (patch column k):  rows    u0 = d0 - d2, u1 = d1 + d2, u2 = d2 - d1, u3 = d1 - d3
    //   steps 12-15 (patch row q):     columns v0 = c0 - c2, v1 = c1 + c2, v2 = c2 - c1, v3 = c1 - c3
    // Schedule over the 16 MFMA gaps of a block: gaps 0-3 carry the k-step's filter requests, gaps 4-7 the LDS reads of
    // the next block's B operands, so the vector work is spread 2, 2, 2, 2 | 2 x 8 | 4, 4, 4, 4 (scalar form).
    float tc[4][4], tu[4][4];         // scalar form: [patch column][patch row]
    f32x2 tA[4], tB[4], tuA[4], tuB[4];   // packed form: A = (x0, x1), B = (right, left) per patch row
    auto tr_step = [&](int sg, int s, float (&v)[16]) __attribute__((always_inline)) {
        if constexpr (PK) {
            if (s < 4) {
                const int q = s;
                tA[q] = pp[sg][q];
                tB[q][0] = dpp_from_right(pe[sg][q][0], tA[q][0]);
                tB[q][1] = dpp_from_left(pe[sg][q][1], tA[q][1]);
            } else if (s == 4) tuA[0] = pk_sub(tA[0], tA[2]);
            else if (s == 5) tuA[1] = pk_add(tA[1], tA[2]);
            else if (s == 6) tuA[2] = pk_sub(tA[2], tA[1]);
            else if (s == 7) tuA[3] = pk_sub(tA[1], tA[3]);
            else if (s == 8) tuB[0] = pk_sub(tB[0], tB[2]);
            else if (s == 9) tuB[1] = pk_add(tB[1], tB[2]);
            else if (s == 10) tuB[2] = pk_sub(tB[2], tB[1]);
            else if (s == 11) tuB[3] = pk_sub(tB[1], tB[3]);
            else {
                const int q = s - 12;
                f32x2 v30, v12;     // (v3, v0) = (A.x - B.x, B.y - A.y);  (v1, v2) = (A.x + A.y, A.y - A.x)
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v30) : "v"(tuA[q]), "v"(tuB[q]));
                asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(v12) : "v"(tuA[q]));
                v[4 * q] = v30[1]; v[4 * q + 1] = v12[0]; v[4 * q + 2] = v12[1]; v[4 * q + 3] = v30[0];
            }
        } else {
            if (s < 4) {
                const int q = s;
                tc[1][q] = pp[sg][q][0]; tc[2][q] = pp[sg][q][1];
                tc[3][q] = dpp_from_right(pe[sg][q][0], tc[1][q]);
                tc[0][q] = dpp_from_left(pe[sg][q][1], tc[2][q]);
            } else if (s < 12) {
                const int k = (s - 4) >> 1;
                if (((s - 4) & 1) == 0) { tu[k][0] = tn_sub(tc[k][0], tc[k][2]); tu[k][1] = tn_add(tc[k][1], tc[k][2]); }
                else { tu[k][2] = tn_sub(tc[k][2], tc[k][1]); tu[k][3] = tn_sub(tc[k][1], tc[k][3]); }
            } else {
                const int q = s - 12;
                v[4 * q] = tn_sub(tu[0][q], tu[2][q]); v[4 * q + 1] = tn_add(tu[1][q], tu[2][q]);
                v[4 * q + 2] = tn_sub(tu[2][q], tu[1][q]); v[4 * q + 3] = tn_sub(tu[1][q], tu[3][q]);
            }
        }
    };
    auto transform = [&](int sg, float (&v)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) tr_step(sg, s, v);
    };
    // ring addressing: one lane base per slot, everything else is an immediate offset
    f32x4* const rb0 = ring + lane;
    auto put = [&](f32x4* slot, int i, const float (&vv)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 tq = {vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]};
            slot[((wave * NB + i) * 4 + q) * 64] = tq;
        }
    };
    auto get = [&](const f32x4* slot, int st, int i, f32x4 (&bb)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[q] = slot[((st * NB + i) * 4 + q) * 64];
    };

    constexpr int NKS = 32, NIT = NKS / 4;
    float vt[16];
#ifdef WN_PROF2
    unsigned long long t_data = 0, t_puts = 0;
#endif
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        transform(i, vt);
#ifdef WN_PROF2
        if (i == 0) { asm volatile("" : "+v"(vt[0])); __builtin_amdgcn_sched_barrier(0); t_data = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
        put(rb0, i, vt);
        load_patch(i, 4 + wave);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef WN_PROF2
    t_puts = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    __syncthreads();
    f32x4 bq[2][4];
    get(rb0, 0, 0, bq[0]);
#ifdef WN_PROF
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef WN_PROF3
    const unsigned long long r_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    // slot rotation: rd = slot read by this iteration, wr = slot written (and read by the next one), fr3 = the third.
    //
    // The loop body is written in its issue order and pinned (sched_barrier after every MFMA and the instructions tucked
    // behind it): with one wave per SIMD nothing else fills the matrix pipe's issue gaps, so which memory / LDS / vector
    // instruction follows which MFMA is the schedule.  The MFMA itself is inline asm with the accumulator as a tied "+a"
    // operand: the builtin form let the register allocator untie destination and source across the loop back edge for the
    // last k-step's accumulators (130-250 v_accvgpr moves per iteration, measured in the ISA).  Its operands come from
    // buffer loads and LDS reads only -- the compiler counts those and places the waits; no VALU -> MFMA hazard is hidden
    // from it.
    f32x4* rd = rb0; f32x4* wr = rb0 + SLOT; f32x4* fr3 = rb0 + 2 * SLOT;
    for (int j = 0; j < NIT; ++j) {
        const int kp = 4 * (j + 2) + wave < NKS ? 4 * (j + 2) + wave : NKS - 4 + wave;     // own k-step two iterations on
        const int kf = 4 * j + TN_FST - 1;                                                 // filter k-step requested at st = 0
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                constexpr int LAST = 4 * NB - 1;
                const int blk = st * NB + i;
                // own work of this block: segment st is transformed in block (st, 0), written to the ring and its patch
                // registers re-requested (own k-step of iteration j + 2) one block later
                const bool do_tr = st < NB && i == 0;
                const bool do_put = NB == 1 ? (st == 1) : (st < NB && i == 1);
                const int tr_seg = NB == 1 ? 0 : st;
                const int fs = (st + TN_FST - 1) % TN_FST;                                  // filter slot freed by k-step st - 1
                const int fso = (ct * 32 + (kf + st < NKS ? kf + st : NKS - 1)) * 4096;
                if (blk == LAST && !(TN_ABL & 32)) {
                    // every wave has written its k-step of the next slot; the slot this iteration reads stays untouched until
                    // the barrier of the NEXT iteration has been passed
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4* nsl = blk < LAST ? rd + ((blk + 1) * 4) * 64 : wr;             // B operands of the next block
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][p]) : "v"(fl[st][p >> 2][p & 3]), "v"(bq[blk & 1][p >> 2][p & 3]));
#if TN_SCHED == 0
                    // spread: one memory / LDS instruction and 2-4 vector instructions behind each MFMA
                    if (i == 0 && p < 4 && !(TN_ABL & 4))                                   // the k-step's filter request
                        fl[fs][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + p * 1024u, fso, 0));
                    if (p >= 4 && p < 8 && !(TN_ABL & 8)) bq[(blk + 1) & 1][p - 4] = nsl[(p - 4) * 64];     // 1 LDS read
                    if (do_tr && !(TN_ABL & 1) && p >= TN_TR_AT && p < TN_TR_AT + TN_TR_N) {
#pragma unroll
                        for (int s2 = (p - TN_TR_AT) * (16 / TN_TR_N); s2 < (p - TN_TR_AT + 1) * (16 / TN_TR_N); ++s2) tr_step(tr_seg, s2, vt);
                    }
                    if (do_put && p >= 4 && p < 8 && !(TN_ABL & 16)) {                      // 1 LDS write
                        const int q = p - 4;
                        const f32x4 tq = {vt[4 * q], vt[4 * q + 1], vt[4 * q + 2], vt[4 * q + 3]};
                        wr[((wave * NB + tr_seg) * 4 + q) * 64] = tq;
                    }
                    if (do_put && p >= 8 && !(TN_ABL & 2)) {                                // patch re-request, 1 load per MFMA
                        const int q = (p - 8) >> 1, so = kp * 4 * HW * 4;
                        if ((p & 1) == 0) pp[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], o0[tr_seg][q], so, 0));
                        else pe[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], oe[tr_seg][q], so, 0));
                    }
#else
                    // clustered: everything that is not an MFMA goes into a few gaps as bursts (the first extra instruction in a
                    // gap between two MFMAs is the expensive one, MI355X_MICROARCH.md "one EXTRA issue slot")
                    if (i == 0 && p == TN_P_FILT && !(TN_ABL & 4)) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            fl[fs][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + q * 1024u, fso, 0));
                    }
                    if (p == TN_P_READ && !(TN_ABL & 8)) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[(blk + 1) & 1][q] = nsl[q * 64];
                    }
                    if (do_tr && !(TN_ABL & 1)) {
#pragma unroll
                        for (int s2 = 0; s2 < 16; ++s2)
                            if (TN_TR_GAPS == 1 ? p == TN_P_TR : (TN_TR_GAPS == 2 ? p == TN_P_TR + 4 * (s2 >> 3) : p == TN_P_TR + 2 * (s2 >> 2))) tr_step(tr_seg, s2, vt);
                    }
                    if (do_put && p == TN_P_PUT) {
                        if (!(TN_ABL & 16)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 tq = {vt[4 * q], vt[4 * q + 1], vt[4 * q + 2], vt[4 * q + 3]};
                                wr[((wave * NB + tr_seg) * 4 + q) * 64] = tq;
                            }
                        }
                        if (!(TN_ABL & 2)) {
                            const int so = kp * 4 * HW * 4;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                pp[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], o0[tr_seg][q], so, 0));
                                pe[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], oe[tr_seg][q], so, 0));
                            }
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        f32x4* const t = rd; rd = wr; wr = fr3; fr3 = t;
    }
    // ---- At M A, BN fold, activation, residuals, store: lane (kq, tile tj) holds channels 16 ct + 4 kq + r of its tile ----
    // Like the prologue this is serial time of the whole CU (no second work-group to overlap with), so the residual requests
    // go out FIRST -- before the wait for the last MFMAs -- and the output transform of every accumulator runs under their
    // latency; only the final adds wait for them.
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    const f32x4 sc4 = *(const f32x4*)(a.scale + 16 * ct + 4 * kq);
    const f32x4 sh4 = *(const f32x4*)(a.shift + 16 * ct + 4 * kq);
    const int img_bytes = WN_C * HW * 4;
    __amdgpu_buffer_rsrc_t yr[NB];
    unsigned lo0[NB], lo1[NB];
    f32x2 ra0[NB][4], ra1[NB][4], rb0v[NB][4], rb1v[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const size_t ib = (size_t)sg_n[i] * WN_C * HW;
        yr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + ib), 0, img_bytes, 0x00020000);
        // an absent residual is a descriptor of zero records: every load returns 0, no per-channel control flow
        const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res1 ? a.res1 + ib : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res2 ? a.res2 + ib : a.x), 0, a.res2 ? img_bytes : 0, 0x00020000);
        const int oy = 2 * sg_ty[i], ox = 2 * (sg_gx[i] * 16 + tj);
        const bool inside = sg_ok[i] && oy < H && ox < W;
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
    __builtin_amdgcn_sched_barrier(0);
    // The last MFMAs' results are read by compiler code below, and inline asm is opaque to the hazard recogniser: left
    // alone it schedules v_accvgpr_read right behind the last MFMA with one wait state (measured: wrong outputs in the
    // NB = 2 build).  Volatile asm statements keep their order, and every accumulator passes through an empty one AFTER
    // the pad, so no read of an accumulator can be scheduled above it.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[i][p]));
#ifdef WN_PROF
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef WN_PROF3
    const unsigned long long r_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
    f32x2 q0[NB][4], q1[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0[4], t1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[i][c][r], m1 = acc[i][4 + c][r], m2 = acc[i][8 + c][r], m3 = acc[i][12 + c][r];
                t0[c] = m0 + m1 + m2;
                t1[c] = m1 - m2 - m3;
            }
            float o00 = t0[0] + t0[1] + t0[2], o01 = t0[1] - t0[2] - t0[3];
            float o10 = t1[0] + t1[1] + t1[2], o11 = t1[1] - t1[2] - t1[3];
            o00 = fmaf(o00, sc4[r], sh4[r]); o01 = fmaf(o01, sc4[r], sh4[r]);
            o10 = fmaf(o10, sc4[r], sh4[r]); o11 = fmaf(o11, sc4[r], sh4[r]);
            o00 = fmaxf(o00, relu_lo); o01 = fmaxf(o01, relu_lo); o10 = fmaxf(o10, relu_lo); o11 = fmaxf(o11, relu_lo);
            q0[i][r] = f32x2{o00, o01}; q1[i][r] = f32x2{o10, o11};
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { asm volatile("" : "+v"(q0[i][r])); asm volatile("" : "+v"(q1[i][r])); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 v0 = q0[i][r] + ra0[i][r], v1 = q1[i][r] + ra1[i][r];
            v0 += rb0v[i][r]; v1 += rb1v[i][r];
            const int so = (16 * ct + r) * HW * 4;
            // WT: single-round launches store write-through (sc1): nothing is left dirty in the L2s for the kernel boundary to
            // write back, which shortens the gap to the next layer's launch (Kodak layer 33.5 -> 32.5 us, A/B'd); launches
            // of several rounds keep plain stores (4K map: write-through 1.5 % slower -- later rounds re-read their neighbours'
            // rows from L2).
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v0), yr[i], lo0[i], so, WT ? 16 : 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v1), yr[i], lo1[i], so, WT ? 16 : 0);
        }
    }
#ifdef WN_PROF
    if (a.prof && lane == 0) {
        unsigned long long* d = a.prof + 4 * ((size_t)blockIdx.x * 4 + wave);
#ifdef WN_PROF3
        d[0] = __builtin_amdgcn_s_memtime() - t_entry; d[1] = __builtin_amdgcn_s_memrealtime() - r_entry;
        d[2] = t_loop1 - t_loop0; d[3] = r_loop1 - r_loop0;
#elif defined(WN_PROF2)
        // prologue split: address set-up | first patch data transformed | all ring writes issued | barrier + first operands
        d[0] = t_addr - t_entry; d[1] = t_data - t_addr; d[2] = t_puts - t_data; d[3] = t_loop0 - t_puts;
#else
        d[0] = t_loop0 - t_entry; d[1] = t_loop1 - t_loop0; d[2] = __builtin_amdgcn_s_memtime() - t_loop1; d[3] = t_entry;
#endif
    }
#endif
}

int icx_wino_tn_launch(const WnArgs& a_in, int nb, int scalar_transform, hipStream_t st) {
    if (a_in.ngroups <= 0) return IC_OK;
    if (nb < 1 || nb > 3 || (a_in.W & 1)) return IC_ERR_UNSUPPORTED;
    const unsigned jobs = (unsigned)((2 * a_in.ngroups + nb - 1) / nb);
    const dim3 grid(2 * jobs), block(256);
    WnArgs a = a_in;
    const bool wt = 2 * jobs <= (nb == 1 ? 512u : 256u);            // everything resident at once: one round
    a.store_wt = wt ? 1 : 0;
    // reciprocals for the kernel's two divisions: n / d = mulhi(n, 2^32 / d + 1) exactly while n d < 2^32
    if ((unsigned long long)(a.g0 + a.ngroups) * (unsigned)(a.gcols > a.grows ? a.gcols : a.grows) >= (1ull << 32)) return IC_ERR_UNSUPPORTED;
    a.mg_cols = a.gcols > 1 ? (unsigned)((1ull << 32) / (unsigned)a.gcols) + 1u : 0u;
    a.mg_rows = a.grows > 1 ? (unsigned)((1ull << 32) / (unsigned)a.grows) + 1u : 0u;
#define TN_GO(NB_, PK_, WT_) hipLaunchKernelGGL((wino3x3_c128_tn_kernel<NB_, PK_, WT_>), grid, block, 0, st, a)
    if (scalar_transform) {
        if (wt) { if (nb == 1) TN_GO(1, false, true); else if (nb == 2) TN_GO(2, false, true); else TN_GO(3, false, true); }
        else { if (nb == 1) TN_GO(1, false, false); else if (nb == 2) TN_GO(2, false, false); else TN_GO(3, false, false); }
    } else {                        // the packed-transform form is a measurement variant: plain stores only
        if (nb == 1) TN_GO(1, true, false); else if (nb == 2) TN_GO(2, true, false); else TN_GO(3, true, false);
    }
#undef TN_GO
    IC_LAUNCH_CHECK();
    return IC_OK;
}
