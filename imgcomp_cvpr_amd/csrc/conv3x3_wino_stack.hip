// The residual stack of the autoencoder (autoencoder.py:224-234 / :252-262, 6B + 2 convs of 3x3 128 -> 128 + BN [+ ReLU] + skips)
// as ONE persistent launch of the NB-segment Winograd kernel (conv3x3_wino_tn.hip) -- for maps whose NB-segment jobs fit the
// chip in one round (a Kodak map: 256 work-groups, one per CU; a 64 x 64 map: 128 work-groups).
//
// Why.  One layer of a Kodak map is 32 us, of which 20.5 us is matrix-pipe time; 2.2-3.3 us is the kernel boundary (drain,
// dispatch, fill) and 3.3 us the serial head and tail of the single resident work-group -- on a 64 x 64 map (a 256 x 256
// image) boundary + head + tail are HALF of a 13 us launch.  Here a work-group keeps its job -- NB segments x one half of the
// output channels -- through all layers:
//   * the geometry (segment decode, lane offsets, neighbour list) is computed once;
//   * a layer boundary is a neighbour hand-off, not a grid barrier: a job's patches at layer l + 1 cover its own segments and
//     the eight segments around each of them, so the work-group waits for the (at most 54) work-groups that own those --
//     both channel halves -- and nobody else.  Flag f[b] = number of layers work-group b has completed; outputs are stored
//     write-through (sc1), drained (vmcnt(0)), then the flag is stored (sc1); readers poll with sc1 loads and read
//     activations with sc1 loads (per-XCD L2s are not coherent for plain accesses; MI355X_MICROARCH.md, inter-workgroup
//     visibility, form R1).  A neighbour can be at most one layer apart, and a layer never writes the buffer it reads, so
//     "neighbours completed layer l" covers write-after-read on the ping-pong buffers as well;
//   * the next layer's first filter fragments are requested behind the output stores and land under their drain.
// Same operations per output as the per-layer kernel: results are bit-identical (tested against it).
//
// All work-groups must be resident at once (they wait for each other): the launcher only takes shapes with <= 256 work-groups
// of one per CU (NB >= 2) or <= 512 of two per CU (NB = 1).  Every spin is bounded: on time-out the kernel raises the error
// word in front of the flags and runs to completion without waiting (wrong results; the word is readable through
// ic_ae_sync_pos_bytes / ic_ae_res_stack_sync_pos_bytes, and the tests check it).
#include "wino_common.h"
#include "internal.h"

#define TS_FST 4
#define TS_TR_AT 12
#define TS_TR_N 4
#define SC1 16                    // aux bits of the raw buffer builtins: sc1 = agent scope (bypasses the CU's L1 / write-through)

__device__ __forceinline__ float ts_add(float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float ts_sub(float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }

template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void wino3x3_c128_stack_kernel(const WnStackArgs a) {
    constexpr int SLOT = 4 * NB * 4 * 64;
    __shared__ f32x4 ring[3 * SLOT];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int hc = b & 1, job = b >> 1;
    const int ct = 4 * hc + wave;
    const int H = a.H, W = a.W, HW = H * W;
    const int nseg = 2 * a.ngroups;

    // ---- the job's segments (fixed for the whole stack): segment s = tile row (s & 1) of tile group s / 2 ----
    int sg_n[NB], sg_ty[NB], sg_gx[NB];
    bool sg_ok[NB];
    {
        const int s0 = job * NB;
        const int g = s0 >> 1;
        const int t = a.mg_cols ? (int)__umulhi((unsigned)g, a.mg_cols) : g;
        int gx = g - t * a.gcols;
        int n = a.mg_rows ? (int)__umulhi((unsigned)t, a.mg_rows) : t;
        int gy = t - n * a.grows, r = s0 & 1;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            sg_ok[i] = s0 + i < nseg;
            sg_n[i] = sg_ok[i] ? n : 0; sg_ty[i] = 2 * gy + r; sg_gx[i] = gx;
            r ^= 1;
            gx += r == 0;
            const bool wx = gx == a.gcols; gx = wx ? 0 : gx;
            gy += wx;
            const bool wy = gy == a.grows; gy = wy ? 0 : gy;
            n += wy;
        }
    }
    // Lane offsets of the patch rows.  They are live through the k-loop (patch re-requests) but NOT through the epilogue, whose
    // residual operands need the registers: they are recomputed per layer behind the output stores, from a per-layer opaque
    // copy of the lane id (otherwise the compiler hoists them out of the layer loop and spills the epilogue's loads).
    unsigned o0[NB][4], oe[NB][4];
    auto patch_offsets = [&](int lane_o) __attribute__((always_inline)) {
        const int kq_o = lane_o >> 4, tj_o = lane_o & 15;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int tx = sg_gx[i] * 16 + tj_o;
            const int r0 = 2 * sg_ty[i] - 1;
            const int ecol = 2 * tx + (tj_o == 0 ? -2 : 2);
            const bool has_e = (tj_o == 0 || tj_o == 15) && ecol >= 0 && ecol < W;
            const bool has_0 = 2 * tx < W;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + q;
                const bool rok = sg_ok[i] && r >= 0 && r < H;
                const unsigned rb = (unsigned)(kq_o * HW + r * W) * 4u;
                o0[i][q] = (rok && has_0) ? rb + 8u * tx : WN_OOB;
                oe[i][q] = (rok && has_e) ? rb + 4u * ecol : WN_OOB;
            }
        }
    };
    patch_offsets(lane);
    // ---- neighbour list (wave 0, one candidate per lane): segment i, row offset dy, column offset dx, channel half ----
    // the owner of segment s' is job s' / NB; its two work-groups are 2 (s' / NB) + {0, 1} in the job order b counts in
    int cand = -1;
    if (wave == 0) {
        const int i = lane / 18, rem = lane - 18 * i, d9 = rem >> 1, half = rem & 1;
        const int dy = d9 / 3 - 1, dx = d9 - 3 * (d9 / 3) - 1;
        int cty = 0, cgx = 0, cn = 0;
        bool cok = false;
#pragma unroll
        for (int ii = 0; ii < NB; ++ii)
            if (i == ii) { cty = sg_ty[ii]; cgx = sg_gx[ii]; cn = sg_n[ii]; cok = sg_ok[ii]; }
        const int ty = cty + dy, gx = cgx + dx;
        if (cok && ty >= 0 && ty < 2 * a.grows && gx >= 0 && gx < a.gcols) {
            const int s2 = 2 * ((cn * a.grows + (ty >> 1)) * a.gcols + gx) + (ty & 1);
            cand = 2 * (s2 / NB) + half;
        }
    }
    unsigned* const flags = a.flags + 1;                          // word 0 of the sync area is the time-out word

    f32x2 pp[NB][4], pe[NB][4];
    __amdgpu_buffer_rsrc_t xr[NB];
    const unsigned fo = lane * 16u;
    f32x4 fl[TS_FST][4];
    f32x4 acc[NB][16];
    float tc[4][4], tu[4][4];
    float vt[16];
    f32x4* const rb0 = ring + lane;
    const int img_bytes = WN_C * HW * 4;

    auto tr_step = [&](int sg, int s, float (&v)[16]) __attribute__((always_inline)) {
        if (s < 4) {
            const int q = s;
            tc[1][q] = pp[sg][q][0]; tc[2][q] = pp[sg][q][1];
            tc[3][q] = dpp_from_right(pe[sg][q][0], tc[1][q]);
            tc[0][q] = dpp_from_left(pe[sg][q][1], tc[2][q]);
        } else if (s < 12) {
            const int k = (s - 4) >> 1;
            if (((s - 4) & 1) == 0) { tu[k][0] = ts_sub(tc[k][0], tc[k][2]); tu[k][1] = ts_add(tc[k][1], tc[k][2]); }
            else { tu[k][2] = ts_sub(tc[k][2], tc[k][1]); tu[k][3] = ts_sub(tc[k][1], tc[k][3]); }
        } else {
            const int q = s - 12;
            v[4 * q] = ts_sub(tu[0][q], tu[2][q]); v[4 * q + 1] = ts_add(tu[1][q], tu[2][q]);
            v[4 * q + 2] = ts_sub(tu[2][q], tu[1][q]); v[4 * q + 3] = ts_sub(tu[1][q], tu[3][q]);
        }
    };

    // filter fragments of layer 0, k-steps 0 .. 2
    {
        const __amdgpu_buffer_rsrc_t fr0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.layers[0].wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int st = 0; st < TS_FST - 1; ++st)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                fl[st][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr0, fo + q * 1024u, (ct * 32 + st) * 4096, 0));
    }

    const int nlayers = a.nlayers;
    for (int l = 0; l < nlayers; ++l) {
        const WnStackLayer& L = a.layers[l];
        // ---- patches of the own k-step of ring slot 0 (sc1: the producer is another CU, possibly another XCD) ----
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            xr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(L.x + (size_t)sg_n[i] * WN_C * HW), 0, img_bytes, 0x00020000);
            const int so = wave * 4 * HW * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], o0[i][q], so, SC1));
                pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], oe[i][q], so, SC1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(L.wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
        // accumulators cleared under the request latency
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

        constexpr int NKS = 32, NIT = NKS / 4;
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
                pp[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], o0[i][q], so, SC1));
                pe[i][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[i], oe[i][q], so, SC1));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        f32x4 bq[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[0][q] = rb0[q * 64];

        // ---- k-loop: the issue order of conv3x3_wino_tn.hip (TN_SCHED 0) ----
        f32x4* rd = rb0; f32x4* wr = rb0 + SLOT; f32x4* fr3 = rb0 + 2 * SLOT;
        for (int j = 0; j < NIT; ++j) {
            const int kp = 4 * (j + 2) + wave < NKS ? 4 * (j + 2) + wave : NKS - 4 + wave;
            const int kf = 4 * j + TS_FST - 1;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    constexpr int LAST = 4 * NB - 1;
                    const int blk = st * NB + i;
                    const bool do_tr = st < NB && i == 0;
                    const bool do_put = NB == 1 ? (st == 1) : (st < NB && i == 1);
                    const int tr_seg = NB == 1 ? 0 : st;
                    const int fs = (st + TS_FST - 1) % TS_FST;
                    const int fso = (ct * 32 + (kf + st < NKS ? kf + st : NKS - 1)) * 4096;
                    if (blk == LAST) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f32x4* nsl = blk < LAST ? rd + ((blk + 1) * 4) * 64 : wr;
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][p]) : "v"(fl[st][p >> 2][p & 3]), "v"(bq[blk & 1][p >> 2][p & 3]));
                        if (i == 0 && p < 4)
                            fl[fs][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo + p * 1024u, fso, 0));
                        if (p >= 4 && p < 8) bq[(blk + 1) & 1][p - 4] = nsl[(p - 4) * 64];
                        if (do_tr && p >= TS_TR_AT && p < TS_TR_AT + TS_TR_N) {
#pragma unroll
                            for (int s2 = (p - TS_TR_AT) * (16 / TS_TR_N); s2 < (p - TS_TR_AT + 1) * (16 / TS_TR_N); ++s2) tr_step(tr_seg, s2, vt);
                        }
                        if (do_put && p >= 4 && p < 8) {
                            const int q = p - 4;
                            const f32x4 tq = {vt[4 * q], vt[4 * q + 1], vt[4 * q + 2], vt[4 * q + 3]};
                            wr[((wave * NB + tr_seg) * 4 + q) * 64] = tq;
                        }
                        if (do_put && p >= 8) {
                            const int q = (p - 8) >> 1, so = kp * 4 * HW * 4;
                            if ((p & 1) == 0) pp[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], o0[tr_seg][q], so, SC1));
                            else pe[tr_seg][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr[tr_seg], oe[tr_seg][q], so, SC1));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            f32x4* const t = rd; rd = wr; wr = fr3; fr3 = t;
        }

        // ---- epilogue: residual requests first, then At M A + BN + act + adds ----
        const float relu_lo = L.relu ? 0.f : -__builtin_inff();
        const f32x4 sc4 = *(const f32x4*)(L.scale + 16 * ct + 4 * kq);
        const f32x4 sh4 = *(const f32x4*)(L.shift + 16 * ct + 4 * kq);
        __amdgpu_buffer_rsrc_t yr[NB];
        f32x2 ra0[NB][4], ra1[NB][4], rb0v[NB][4], rb1v[NB][4];
        unsigned lo0[NB], lo1[NB];
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int oy = 2 * sg_ty[i], ox = 2 * (sg_gx[i] * 16 + (lane_e & 15));
            const bool inside = sg_ok[i] && oy < H && ox < W;
            lo0[i] = inside ? (unsigned)((4 * (lane_e >> 4) * HW + oy * W + ox) * 4) : WN_OOB;
            lo1[i] = inside && oy + 1 < H ? lo0[i] + 4u * W : WN_OOB;
            const size_t ib = (size_t)sg_n[i] * WN_C * HW;
            yr[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(L.y + ib), 0, img_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc((void*)(L.res1 ? L.res1 + ib : L.x), 0, L.res1 ? img_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc((void*)(L.res2 ? L.res2 + ib : L.x), 0, L.res2 ? img_bytes : 0, 0x00020000);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int so = (16 * ct + r) * HW * 4;
                ra0[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo0[i], so, SC1));
                ra1[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r1r, lo1[i], so, SC1));
                rb0v[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo0[i], so, SC1));
                rb1v[i][r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r2r, lo1[i], so, SC1));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[i][p]));
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
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v0), yr[i], lo0[i], so, SC1);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v1), yr[i], lo1[i], so, SC1);
            }
        }
        if (l + 1 == nlayers) break;
        // the next layer's first filter fragments: they depend on nobody, and their latency passes under the store drain
        __builtin_amdgcn_sched_barrier(0);
        {
            const __amdgpu_buffer_rsrc_t frn = __builtin_amdgcn_make_buffer_rsrc((void*)(a.layers[l + 1].wp + WN_FRAG_FLOATS), 0, WN_FRAG_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int st = 0; st < TS_FST - 1; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    fl[st][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frn, fo + q * 1024u, (ct * 32 + st) * 4096, 0));
        }
        {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            patch_offsets(lane_o);
        }

        // ---- hand-off: drain the write-through stores, publish "l + 1 layers done", wait for the neighbours ----
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                   // the flag store stays behind the barrier (s_barrier is not a compiler fence)
        if (wave == 0) {
            if (lane == 0) __hip_atomic_store(flags + b, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned need = (unsigned)(l + 1);
            bool ok = cand < 0;
            unsigned spins = 0;
            while (true) {
                if (!ok) ok = __hip_atomic_load(flags + cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
                // lane 63 never has a candidate: it watches the time-out word, so that one time-out releases everybody
                const bool dead = lane == 63 && __hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0 || __builtin_amdgcn_ballot_w64(dead) != 0) break;
                if (++spins > a.spin_limit) {
                    if (lane == 0) __hip_atomic_store(a.flags, 1u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                   // ... and the next layer's loads behind this one
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Can this shape run as one persistent launch?  nb = segments per job the per-layer plan would use.
bool icx_wino_stack_fits(int N, int H, int W, int nb, int nlayers) {
    if (nb < 1 || nb > 3 || (W & 1) || nlayers < 1 || nlayers > WN_STACK_MAX_LAYERS) return false;
    if ((long long)WN_C * H * W * 4 >= (1ll << 31)) return false;
    const long long groups = (long long)N * ic_cdiv(H, 4) * ic_cdiv(W, 32);
    const long long wgs = 2 * ((2 * groups + nb - 1) / nb);
    return wgs <= (nb == 1 ? 512 : 256);
}

// a.flags: the sync area, WN_STACK_SYNC_BYTES (time-out word, then one flag per work-group); zeroed here, on the stream
int icx_wino_stack_launch(WnStackArgs& a, int nb, hipStream_t st) {
    const int groups = a.N * a.grows * a.gcols;
    if (!icx_wino_stack_fits(a.N, a.H, a.W, nb, a.nlayers)) return IC_ERR_UNSUPPORTED;
    const unsigned jobs = (unsigned)((2 * groups + nb - 1) / nb);
    a.ngroups = groups;
    if ((unsigned long long)groups * (unsigned)(a.gcols > a.grows ? a.gcols : a.grows) >= (1ull << 32)) return IC_ERR_UNSUPPORTED;
    a.mg_cols = a.gcols > 1 ? (unsigned)((1ull << 32) / (unsigned)a.gcols) + 1u : 0u;
    a.mg_rows = a.grows > 1 ? (unsigned)((1ull << 32) / (unsigned)a.grows) + 1u : 0u;
    if (!a.spin_limit) a.spin_limit = 1u << 20;                 // ~ a second of polling: far beyond any real skew
    hipError_t e = hipMemsetAsync(a.flags, 0, (2 * jobs + 1) * sizeof(unsigned), st);            // <= 513 words of WN_STACK_SYNC_BYTES
    if (e != hipSuccess) return (int)e;
    const dim3 grid(2 * jobs), block(256);
    if (nb == 1) hipLaunchKernelGGL(wino3x3_c128_stack_kernel<1>, grid, block, 0, st, a);
    else if (nb == 2) hipLaunchKernelGGL(wino3x3_c128_stack_kernel<2>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(wino3x3_c128_stack_kernel<3>, grid, block, 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
