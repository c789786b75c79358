// Filter gradient of the 3x3 128 -> 128 layer in the Winograd domain (training; reference: the tf.gradients of
// slim.conv2d in code/autoencoder.py:274-287 w.r.t. its `weights`).
//
//   forward   Y = At [ sum_ci (G g Gt) (.) (Bt d B) ] A                                  (conv3x3_wino.hip)
//   gradient  dL/dg[ci][co] = Gt [ sum_tiles (Bt d B)[ci] (.) (A dY At)[co] ] G
//
// i.e. 16 independent GEMMs  dG^[pos] (128 ci x 128 co) = V[pos] (ci x tiles) . U[pos]^T (tiles x co)  over ALL 2x2 tiles of
// the batch -- 16/36 of the multiply-adds of the direct form (conv_wgrad.hip), whose 9 taps re-read the same operands.
// V is the forward layer's input transform of x, U the 2x2 -> 4x4 expansion of the output gradient.
//
// Work-group = 64 ci x 64 co x 16 positions over a K-slice of tiles; 4 waves = 2 x 2 blocks of 32 x 32, 16 accumulators of
// 16 registers (all 256 AGPRs), v_mfma_f32_32x32x2_f32 with K = 2 tiles per instruction: lane (c = l & 31, t = l >> 5) supplies
// V[pos][ci = c][tile t] and U[pos][co = c][tile t] -- ONE (channel, tile) pair, exactly what one transform produces.
// The transforms want lanes along x (coalesced pixel pairs, neighbours' columns through DPP), the MFMA wants lanes along
// channels, so the transformed values cross through LDS: a chunk of 8 tiles x 128 channels x 16 positions (64 KB) is produced
// while the previous chunk is multiplied (double buffer, one barrier per chunk).  A record = the 16 positions of one
// (tile, channel) = four float4, XOR-swizzled by (channel >> 2) & 3 so that both the producers' stores and the consumers'
// loads (lanes = 32 consecutive channels, 64 B apart) are conflict-free.
// Producers: a wave pass = 8 channels x 8 tiles (lane = channel * 8 + tile): 4 patch rows as aligned pixel pairs, the outer
// patch columns from the neighbour lanes (DPP row shifts; tiles 0 and 7 of a chunk load the pair beyond), zero padding by
// out-of-range buffer offsets.  Per chunk every wave does two V passes and two U passes next to its 64 MFMAs.
// Partial sums of the K-slices go to the workspace already reduced to the 3x3 taps (Gt . G is linear); an ordered reduction
// adds the slices and the weight-decay term (deterministic).
#include "wino_common.h"

#ifndef WW_EIGHT_WAVES
#define WW_EIGHT_WAVES 1              // 1: the 8-wave kernel (two waves per SIMD, 8 positions per wave); 0: the 4-wave form of rounds 2-4
#endif
#define WW_TC 8                       // tiles per chunk
#define WW_TS (128 * 4 + 1)           // float4 slots per tile: 128 records + one slot of padding (see the kernel)

struct WwArgs {
    const float* x; const float* dy; float* partial;
    int N, H, W;
    int tiles_x, tiles_y, chunks_x;   // 2x2 tiles per row / column, 8-tile chunks per tile row
    int nchunks, chunks_per_slice;
};

__device__ __forceinline__ float ww_shr1(float v) {      // lane i <- lane i - 1 within a row of 16 (the row's lane 0: don't care)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float ww_shl1(float v) {      // lane i <- lane i + 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x101, 0xf, 0xf, true));
}
// single adds as inline asm: left to itself the SLP vectoriser packs pairs of them into v_pk_add_f32 and pays for it with
// register shuffles (53 v_mov per chunk in the first build)
__device__ __forceinline__ float ww_add(float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float ww_sub(float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float ww_neg(float x) { float r; asm("v_xor_b32 %0, 0x80000000, %1" : "=v"(r) : "v"(x)); return r; }

__global__ __launch_bounds__(256) void wino3x3_c128_wgrad_kernel(const WwArgs a) {
    // [buffer][tile][channel: 0..63 V (ci), 64..127 U (co)][4 float4, slot j at j ^ ((channel >> 2) & 3)].  Consumers read 32
    // consecutive channels of one tile: the swizzle spreads them over all banks.  Producers write 8 tiles x 8 channels per
    // instruction: tiles are 8 KB apart -- the same banks (measured: the kernel ran at a third of its MFMA rate) -- so every
    // tile is shifted by one more 16-byte slot.
    __shared__ f32x4 lds[2 * WW_TC * WW_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = ic_xcd_run(blockIdx.x, gridDim.x);      // the four channel blocks of a slice read the same tiles: one XCD
    const int blk = b & 3, slice = b >> 2;
    const int cib = blk >> 1, cob = blk & 1;
    const int bi = wave >> 1, bj = wave & 1;
    const int H = a.H, W = a.W, HW = H * W;
    const int ch8 = lane >> 3, tj = lane & 7;
    const int q0 = slice * a.chunks_per_slice;
    const int q1 = q0 + a.chunks_per_slice < a.nchunks ? q0 + a.chunks_per_slice : a.nchunks;
    const unsigned tensor_bytes = (unsigned)a.N * WN_C * HW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, tensor_bytes, 0x00020000);

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- producer side ----
    // Requests: lane part of the byte offset is loop-invariant (channel row ch8, pixel pair of tile tj); everything that
    // moves with the chunk (image, tile row, chunk column) and with the pass (8-channel group) is a scalar offset.
    const unsigned v_own = (unsigned)(ch8 * HW + 2 * tj) * 4u;
    // tile 0: the pair to the left, tile 7: the pair to the right.  The lane offset of a buffer load is unsigned and range-checked
    // on its own, so "- 8 bytes" goes into the descriptor (xre is based 8 bytes before x) and the lane offsets are + 8.
    const unsigned v_edge = tj == 0 ? v_own : v_own + 16u;
    const __amdgpu_buffer_rsrc_t xre = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - 2), 0, tensor_bytes + 8u, 0x00020000);
    f32x2 xv[2][4], xe[2][4], du[2][2];
    // The chunk whose raw operands are being requested: (image, tile row, chunk column) advance incrementally -- no divisions in
    // the loop -- and everything that depends on them is prepared ONCE per chunk: the scalar offsets of the first patch / gradient
    // row and, per patch row, the lane offset with the row's validity already folded in (rows -1 and H.. read as zeros).
    int rq_n, rq_ty, rq_txc, rq_q;
    {
        const int per_img = a.tiles_y * a.chunks_x;
        rq_n = q0 / per_img; const int rem = q0 - rq_n * per_img;
        rq_ty = rem / a.chunks_x; rq_txc = rem - rq_ty * a.chunks_x; rq_q = q0;
    }
    int rq_xbase = 0, rq_gbase = 0;
    unsigned rq_vo[4], rq_ve[4], rq_vg[2];
    auto raw_setup = [&]() __attribute__((always_inline)) {          // prepares chunk rq_q, then steps the coordinates to rq_q + 1
        const bool live = rq_q < q1;
        const int tx = WW_TC * rq_txc + tj;
        const bool tile_ok = live && tx < a.tiles_x;
        const bool e_ok = tile_ok && (tj == 0 ? tx > 0 : (tj == WW_TC - 1 && 2 * tx + 2 < W));
        const unsigned vo = tile_ok ? v_own : WN_OOB, ve = e_ok ? v_edge : WN_OOB;
        const int r0 = 2 * rq_ty - 1;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const bool rok = r0 + r4 >= 0 && r0 + r4 < H;            // uniform
            rq_vo[r4] = rok ? vo : WN_OOB; rq_ve[r4] = rok ? ve : WN_OOB;
        }
        rq_vg[0] = vo; rq_vg[1] = 2 * rq_ty + 1 < H ? vo : WN_OOB;
        const int img = live ? rq_n : 0, row = live ? r0 : 0, col = live ? rq_txc : 0;
        rq_xbase = ((img * WN_C + 64 * cib + 8 * wave) * HW + row * W + 2 * WW_TC * col) * 4;       // patch row 0; may be "row -1": never dereferenced then
        rq_gbase = ((img * WN_C + 64 * cob + 8 * wave) * HW + (row + 1) * W + 2 * WW_TC * col) * 4;
        ++rq_q; ++rq_txc;
        const bool wx = rq_txc == a.chunks_x; rq_txc = wx ? 0 : rq_txc; rq_ty += wx;
        const bool wy = rq_ty == a.tiles_y; rq_ty = wy ? 0 : rq_ty; rq_n += wy;
    };
    // one request of pass p: i = 0..3 own pair of patch row i, 4..7 edge pair of patch row i - 4, 8..9 gradient row i - 8
    auto raw_load = [&](int p, int i) __attribute__((always_inline)) {
        const int pofs = 32 * p * HW * 4;                         // pass 1: channels + 32
        if (i < 4) xv[p][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, rq_vo[i], rq_xbase + pofs + i * W * 4, 0));
        else if (i < 8) xe[p][i - 4] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xre, rq_ve[i - 4], rq_xbase + pofs + (i - 4) * W * 4, 0));
        else du[p][i - 8] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(gr, rq_vg[i - 8], rq_gbase + pofs + (i - 8) * W * 4, 0));
    };
    auto load_raw = [&]() __attribute__((always_inline)) {
        raw_setup();
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 10; ++i) raw_load(p, i);
    };
    // LDS record of (tile tj, channel 8 g + ch8): four float4 slots, slot j at j ^ sw; sw is the same for a wave's four passes
    const int sw = (2 * wave + (ch8 >> 2)) & 3;
    f32x4* const wrec = lds + tj * WW_TS + (8 * wave + ch8) * 4;            // buffer 0, pass 0, V; + 32 channels: pass 1; + 64: U
    int wsl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wsl[j] = j ^ sw;
    // transform state of the pass in flight
    float tc[4][4], tu[4][4];
    // 16 micro-steps of a pass, 3-5 vector instructions each (+ one LDS store in the last eight): placed behind individual MFMAs
    auto micro = [&](int buf, int p, int m) __attribute__((always_inline)) {
        f32x4* const rv = wrec + buf * WW_TC * WW_TS + 32 * p * 4;
        if (m < 4) {                    // patch row m: own pair = columns 1, 2; columns 0 and 3 from the neighbour lanes
            const int r4 = m;
            tc[1][r4] = xv[p][r4][0]; tc[2][r4] = xv[p][r4][1];
            const float right = ww_shl1(tc[1][r4]), left = ww_shr1(tc[2][r4]);
            tc[3][r4] = tj == WW_TC - 1 ? xe[p][r4][0] : right;
            tc[0][r4] = tj == 0 ? xe[p][r4][1] : left;
        } else if (m < 8) {             // Bt d: over the rows of patch column k
            const int k = m - 4;
            tu[k][0] = ww_sub(tc[k][0], tc[k][2]); tu[k][1] = ww_add(tc[k][1], tc[k][2]);
            tu[k][2] = ww_sub(tc[k][2], tc[k][1]); tu[k][3] = ww_sub(tc[k][1], tc[k][3]);
        } else if (m < 12) {            // (Bt d) B: transformed row r4 -> positions 4 r4 .. 4 r4 + 3
            const int r4 = m - 8;
            const f32x4 v = {ww_sub(tu[0][r4], tu[2][r4]), ww_add(tu[1][r4], tu[2][r4]), ww_sub(tu[2][r4], tu[1][r4]), ww_sub(tu[1][r4], tu[3][r4])};
            rv[wsl[r4]] = v;
        } else {                        // U = A dY At, row i of A dY: [p, q] -> [p, p + q, p - q, -q];  A = [[1,0],[1,1],[1,-1],[0,-1]]
            const int i = m - 12;
            const float d00 = du[p][0][0], d01 = du[p][0][1], d10 = du[p][1][0], d11 = du[p][1][1];
            const float pp = i == 0 ? d00 : (i == 1 ? ww_add(d00, d10) : (i == 2 ? ww_sub(d00, d10) : ww_neg(d10)));
            const float qq = i == 0 ? d01 : (i == 1 ? ww_add(d01, d11) : (i == 2 ? ww_sub(d01, d11) : ww_neg(d11)));
            const f32x4 v = {pp, ww_add(pp, qq), ww_sub(pp, qq), ww_neg(qq)};
            (rv + 64 * 4)[wsl[i]] = v;
        }
    };

    // ---- consumer side: operands of a k-step = tiles 2 s, 2 s + 1 of the chunk ----
    const int ca = 32 * bi + (lane & 31), cb = 64 + 32 * bj + (lane & 31);
    const f32x4* ra[4]; const f32x4* rb[4];                               // buffer 0, k-step 0; + 2 WW_TS per k-step
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ra[j] = lds + (lane >> 5) * WW_TS + ca * 4 + (j ^ ((ca >> 2) & 3));
        rb[j] = lds + (lane >> 5) * WW_TS + cb * 4 + (j ^ ((cb >> 2) & 3));
    }
    f32x4 va[2][4], vb[2][4];
    auto fetch = [&](int set, int buf, int s, int j) __attribute__((always_inline)) {      // j = 0..3: A slot j, 4..7: B slot j - 4
        const int o = (buf * WW_TC + 2 * s) * WW_TS;
        if (j < 4) va[set][j] = ra[j][o]; else vb[set][j - 4] = rb[j - 4][o];
    };
    auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (q0 < q1) {
        load_raw();
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int m = 0; m < 16; ++m) micro(0, p, m);
        load_raw();
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) fetch(0, 0, 0, j);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = q0; q < q1; ++q) {
            const int buf = (q - q0) & 1;
            raw_setup();
            // k-steps 0 and 1: the NEXT chunk's two transform passes ride behind the MFMAs (one micro-step per MFMA); operands of
            // k-step s + 1 are fetched behind the first eight MFMAs of k-step s.  Before k-step 3 every wave has written the
            // other buffer: barrier, then k-step 3 fetches the first operands of the next chunk from it.
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int set = s & 1;
                if (s == 3) lds_barrier();
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[set][p >> 2][p & 3], vb[set][p >> 2][p & 3], acc[p], 0, 0, 0);
                    if (p < 8) {
                        if (s < 3) fetch(set ^ 1, buf, s + 1, p); else fetch(0, buf ^ 1, 0, p);
                    }
                    if (s < 2) micro(buf ^ 1, s, p);
                    // raw operands of the chunk after next, each requested right after the micro-steps that consumed its
                    // registers: a whole chunk of MFMAs (1.7 us) covers the round trip
                    if (s < 2 && p >= 4 && p < 12) raw_load(s, p - 4);
                    if ((s == 1 || s == 2) && p < 2) raw_load(s - 1, 8 + p);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- Gt . G over the positions, then the slice's partial dW in the TF layout [ky][kx][ci][co] ----
    float* const out = a.partial + (size_t)slice * 9 * WN_C * WN_C;
    const int co = 64 * cob + 32 * bj + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = 64 * cib + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float h[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d0 = acc[j][r], d1 = acc[4 + j][r], d2 = acc[8 + j][r], d3 = acc[12 + j][r];
            const float s = 0.5f * (d1 + d2);
            h[0][j] = d0 + s; h[1][j] = 0.5f * (d1 - d2); h[2][j] = s + d3;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float s = 0.5f * (h[ky][1] + h[ky][2]);
            const float g0 = h[ky][0] + s, g1 = 0.5f * (h[ky][1] - h[ky][2]), g2 = s + h[ky][3];
            out[((size_t)(ky * 3 + 0) * WN_C + ci) * WN_C + co] = g0;
            out[((size_t)(ky * 3 + 1) * WN_C + ci) * WN_C + co] = g1;
            out[((size_t)(ky * 3 + 2) * WN_C + ci) * WN_C + co] = g2;
        }
    }
}

// ---- 8-wave form (round 5): the same work-group tile (64 ci x 64 co x 16 positions over a K-slice), TWO waves per SIMD ----
// The 4-wave form above holds 16 accumulators of 16 registers per wave -- all 256 AGPRs, one wave per SIMD -- and a lone wave issues in
// order: 57 us per cfg3 layer against a 27 us MFMA floor.  Here waves w and w + 4 share a 32 x 32 channel block and take 8 positions
// each (ph = wave >> 2: the position rows 2 ph, 2 ph + 1 of the 4 x 4 Winograd grid = slots 2 ph, 2 ph + 1 of an LDS record), so a wave
// needs 128 accumulator registers and every SIMD has a second wave to issue from.  The producers' work is split the same way: wave w
// makes ONE transform pass per chunk (p = wave >> 2: channels + 32) instead of two.  LDS layout, K order per accumulator (chunk, k-step,
// tile pair) and therefore every sum are those of the 4-wave form: bit-identical results.  After the loop the ph = 1 waves hand their
// eight accumulators to their partner through the (then free) LDS and the ph = 0 waves apply Gt . G and write the slice's partial.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino3x3_c128_wgrad8_kernel(const WwArgs a) {
    // [buffer][tile][channel: 0..63 V (ci), 64..127 U (co)][4 float4, slot j at j ^ ((channel >> 2) & 3)].  Consumers read 32
    // consecutive channels of one tile: the swizzle spreads them over all banks.  Producers write 8 tiles x 8 channels per
    // instruction: tiles are 8 KB apart -- the same banks (measured: the kernel ran at a third of its MFMA rate) -- so every
    // tile is shifted by one more 16-byte slot.
    __shared__ f32x4 lds[2 * WW_TC * WW_TS];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, ph = wave8 >> 2;      // wave: channel-block role (as in the 4-wave form); ph: position half / transform pass
    const int b = ic_xcd_run(blockIdx.x, gridDim.x);      // the four channel blocks of a slice read the same tiles: one XCD
    const int blk = b & 3, slice = b >> 2;
    const int cib = blk >> 1, cob = blk & 1;
    const int bi = wave >> 1, bj = wave & 1;
    const int H = a.H, W = a.W, HW = H * W;
    const int ch8 = lane >> 3, tj = lane & 7;
    const int q0 = slice * a.chunks_per_slice;
    const int q1 = q0 + a.chunks_per_slice < a.nchunks ? q0 + a.chunks_per_slice : a.nchunks;
    const unsigned tensor_bytes = (unsigned)a.N * WN_C * HW * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, tensor_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, tensor_bytes, 0x00020000);

    f32x16 acc[8];                                          // positions 8 ph .. 8 ph + 7
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- producer side ----
    // Requests: lane part of the byte offset is loop-invariant (channel row ch8, pixel pair of tile tj); everything that
    // moves with the chunk (image, tile row, chunk column) and with the pass (8-channel group) is a scalar offset.
    const unsigned v_own = (unsigned)(ch8 * HW + 2 * tj) * 4u;
    // tile 0: the pair to the left, tile 7: the pair to the right.  The lane offset of a buffer load is unsigned and range-checked
    // on its own, so "- 8 bytes" goes into the descriptor (xre is based 8 bytes before x) and the lane offsets are + 8.
    const unsigned v_edge = tj == 0 ? v_own : v_own + 16u;
    const __amdgpu_buffer_rsrc_t xre = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - 2), 0, tensor_bytes + 8u, 0x00020000);
    f32x2 xv[1][4], xe[1][4], du[1][2];                       // ONE pass per wave: pass ph
    // The chunk whose raw operands are being requested: (image, tile row, chunk column) advance incrementally -- no divisions in
    // the loop -- and everything that depends on them is prepared ONCE per chunk: the scalar offsets of the first patch / gradient
    // row and, per patch row, the lane offset with the row's validity already folded in (rows -1 and H.. read as zeros).
    int rq_n, rq_ty, rq_txc, rq_q;
    {
        const int per_img = a.tiles_y * a.chunks_x;
        rq_n = q0 / per_img; const int rem = q0 - rq_n * per_img;
        rq_ty = rem / a.chunks_x; rq_txc = rem - rq_ty * a.chunks_x; rq_q = q0;
    }
    int rq_xbase = 0, rq_gbase = 0;
    unsigned rq_vo[4], rq_ve[4], rq_vg[2];
    auto raw_setup = [&]() __attribute__((always_inline)) {          // prepares chunk rq_q, then steps the coordinates to rq_q + 1
        const bool live = rq_q < q1;
        const int tx = WW_TC * rq_txc + tj;
        const bool tile_ok = live && tx < a.tiles_x;
        const bool e_ok = tile_ok && (tj == 0 ? tx > 0 : (tj == WW_TC - 1 && 2 * tx + 2 < W));
        const unsigned vo = tile_ok ? v_own : WN_OOB, ve = e_ok ? v_edge : WN_OOB;
        const int r0 = 2 * rq_ty - 1;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const bool rok = r0 + r4 >= 0 && r0 + r4 < H;            // uniform
            rq_vo[r4] = rok ? vo : WN_OOB; rq_ve[r4] = rok ? ve : WN_OOB;
        }
        rq_vg[0] = vo; rq_vg[1] = 2 * rq_ty + 1 < H ? vo : WN_OOB;
        const int img = live ? rq_n : 0, row = live ? r0 : 0, col = live ? rq_txc : 0;
        rq_xbase = ((img * WN_C + 64 * cib + 8 * wave) * HW + row * W + 2 * WW_TC * col) * 4;       // patch row 0; may be "row -1": never dereferenced then
        rq_gbase = ((img * WN_C + 64 * cob + 8 * wave) * HW + (row + 1) * W + 2 * WW_TC * col) * 4;
        ++rq_q; ++rq_txc;
        const bool wx = rq_txc == a.chunks_x; rq_txc = wx ? 0 : rq_txc; rq_ty += wx;
        const bool wy = rq_ty == a.tiles_y; rq_ty = wy ? 0 : rq_ty; rq_n += wy;
    };
    // one request of pass p: i = 0..3 own pair of patch row i, 4..7 edge pair of patch row i - 4, 8..9 gradient row i - 8
    auto raw_load = [&](int p, int i) __attribute__((always_inline)) {
        const int pofs = 32 * ph * HW * 4;                        // pass 1: channels + 32 (p is 0 here: the register set)
        if (i < 4) xv[p][i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, rq_vo[i], rq_xbase + pofs + i * W * 4, 0));
        else if (i < 8) xe[p][i - 4] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xre, rq_ve[i - 4], rq_xbase + pofs + (i - 4) * W * 4, 0));
        else du[p][i - 8] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(gr, rq_vg[i - 8], rq_gbase + pofs + (i - 8) * W * 4, 0));
    };
    auto load_raw = [&]() __attribute__((always_inline)) {
        raw_setup();
#pragma unroll
        for (int i = 0; i < 10; ++i) raw_load(0, i);
    };
    // LDS record of (tile tj, channel 8 g + ch8): four float4 slots, slot j at j ^ sw; sw is the same for a wave's four passes
    const int sw = (2 * wave + (ch8 >> 2)) & 3;
    f32x4* const wrec = lds + tj * WW_TS + (8 * wave + ch8) * 4;            // buffer 0, pass 0, V; + 32 channels: pass 1; + 64: U
    int wsl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wsl[j] = j ^ sw;
    // transform state of the pass in flight
    float tc[4][4], tu[4][4];
    // 16 micro-steps of a pass, 3-5 vector instructions each (+ one LDS store in the last eight): placed behind individual MFMAs
    auto micro = [&](int buf, int p, int m) __attribute__((always_inline)) {
        f32x4* const rv = wrec + buf * WW_TC * WW_TS + 32 * ph * 4;      // p = 0: the register set; the pass is ph
        if (m < 4) {                    // patch row m: own pair = columns 1, 2; columns 0 and 3 from the neighbour lanes
            const int r4 = m;
            tc[1][r4] = xv[p][r4][0]; tc[2][r4] = xv[p][r4][1];
            const float right = ww_shl1(tc[1][r4]), left = ww_shr1(tc[2][r4]);
            tc[3][r4] = tj == WW_TC - 1 ? xe[p][r4][0] : right;
            tc[0][r4] = tj == 0 ? xe[p][r4][1] : left;
        } else if (m < 8) {             // Bt d: over the rows of patch column k
            const int k = m - 4;
            tu[k][0] = ww_sub(tc[k][0], tc[k][2]); tu[k][1] = ww_add(tc[k][1], tc[k][2]);
            tu[k][2] = ww_sub(tc[k][2], tc[k][1]); tu[k][3] = ww_sub(tc[k][1], tc[k][3]);
        } else if (m < 12) {            // (Bt d) B: transformed row r4 -> positions 4 r4 .. 4 r4 + 3
            const int r4 = m - 8;
            const f32x4 v = {ww_sub(tu[0][r4], tu[2][r4]), ww_add(tu[1][r4], tu[2][r4]), ww_sub(tu[2][r4], tu[1][r4]), ww_sub(tu[1][r4], tu[3][r4])};
            rv[wsl[r4]] = v;
        } else {                        // U = A dY At, row i of A dY: [p, q] -> [p, p + q, p - q, -q];  A = [[1,0],[1,1],[1,-1],[0,-1]]
            const int i = m - 12;
            const float d00 = du[p][0][0], d01 = du[p][0][1], d10 = du[p][1][0], d11 = du[p][1][1];
            const float pp = i == 0 ? d00 : (i == 1 ? ww_add(d00, d10) : (i == 2 ? ww_sub(d00, d10) : ww_neg(d10)));
            const float qq = i == 0 ? d01 : (i == 1 ? ww_add(d01, d11) : (i == 2 ? ww_sub(d01, d11) : ww_neg(d11)));
            const f32x4 v = {pp, ww_add(pp, qq), ww_sub(pp, qq), ww_neg(qq)};
            (rv + 64 * 4)[wsl[i]] = v;
        }
    };

    // ---- consumer side: operands of a k-step = tiles 2 s, 2 s + 1 of the chunk ----
    const int ca = 32 * bi + (lane & 31), cb = 64 + 32 * bj + (lane & 31);
    const f32x4* ra[4]; const f32x4* rb[4];                               // buffer 0, k-step 0; + 2 WW_TS per k-step
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ra[j] = lds + (lane >> 5) * WW_TS + ca * 4 + (j ^ ((ca >> 2) & 3));
        rb[j] = lds + (lane >> 5) * WW_TS + cb * 4 + (j ^ ((cb >> 2) & 3));
    }
    f32x4 va[2][2], vb[2][2];
    // j = 0, 1: A slot 2 ph + j; 2, 3: B slot 2 ph + j - 2 (ra / rb hold all four slot pointers; ph is wave-uniform)
    auto fetch = [&](int set, int buf, int s, int j) __attribute__((always_inline)) {
        const int o = (buf * WW_TC + 2 * s) * WW_TS;
        if (j < 2) va[set][j] = (ph ? ra[2 + j] : ra[j])[o]; else vb[set][j - 2] = (ph ? rb[j] : rb[j - 2])[o];
    };
    auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (q0 < q1) {
        load_raw();
#pragma unroll
        for (int m = 0; m < 16; ++m) micro(0, 0, m);
        load_raw();
        lds_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch(0, 0, 0, j);
        __builtin_amdgcn_sched_barrier(0);
        for (int q = q0; q < q1; ++q) {
            const int buf = (q - q0) & 1;
            raw_setup();
            // 8 MFMAs per k-step and wave.  k-steps 0 and 1: the NEXT chunk's transform pass of this wave rides behind them (one
            // micro-step per MFMA: 16 in all); the operands of k-step s + 1 are fetched behind the first four MFMAs of k-step s;
            // the raw operands of the chunk after next are requested in k-steps 2 and 3, after the micro-steps that consumed their
            // registers.  Before k-step 3 every wave has written the other buffer: barrier, then k-step 3 fetches from it.
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int set = s & 1;
                if (s == 3) lds_barrier();
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[set][p >> 2][p & 3], vb[set][p >> 2][p & 3], acc[p], 0, 0, 0);
                    if (p < 4) {
                        if (s < 3) fetch(set ^ 1, buf, s + 1, p); else fetch(0, buf ^ 1, 0, p);
                    }
                    if (s < 2) micro(buf ^ 1, 0, 8 * s + p);
                    if (s == 2) raw_load(0, p);
                    if (s == 3 && p < 2) raw_load(0, 8 + p);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- the ph = 1 waves' accumulators to their partners (LDS is free now: 8 positions x 16 registers x 64 lanes = 32 KB per wave) ----
    __syncthreads();
    {
        float* const xch = reinterpret_cast<float*>(lds) + (size_t)wave * 8 * 16 * 64;
        if (ph == 1) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(p * 16 + r) * 64 + lane] = acc[p][r];
        }
        __syncthreads();
        if (ph == 1) return;
        // (all waves of the work-group have passed both barriers: no barrier below this line)
    // ---- Gt . G over the positions, then the slice's partial dW in the TF layout [ky][kx][ci][co] ----
    float* const out = a.partial + (size_t)slice * 9 * WN_C * WN_C;
    const int co = 64 * cob + 32 * bj + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ci = 64 * cib + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float h[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d0 = acc[j][r], d1 = acc[4 + j][r], d2 = xch[(j * 16 + r) * 64 + lane], d3 = xch[((4 + j) * 16 + r) * 64 + lane];
            const float s = 0.5f * (d1 + d2);
            h[0][j] = d0 + s; h[1][j] = 0.5f * (d1 - d2); h[2][j] = s + d3;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float s = 0.5f * (h[ky][1] + h[ky][2]);
            const float g0 = h[ky][0] + s, g1 = 0.5f * (h[ky][1] - h[ky][2]), g2 = s + h[ky][3];
            out[((size_t)(ky * 3 + 0) * WN_C + ci) * WN_C + co] = g0;
            out[((size_t)(ky * 3 + 1) * WN_C + ci) * WN_C + co] = g1;
            out[((size_t)(ky * 3 + 2) * WN_C + ci) * WN_C + co] = g2;
        }
    }
    }
}

// dW = sum over the slices (+ wd * w): four groups of threads sum a quarter of the slices each, in index order, and the four
// sums are added as (g0 + g1) + (g2 + g3) -- a fixed order, and four times the loads in flight of a sequential loop
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ partial, int S, const float* __restrict__ w,
                                                                float wd, float* __restrict__ dw) {
    __shared__ double red[4][64][4];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e4 = blockIdx.x * 64 + lane;                 // float4 index: 9 * 128 * 128 / 4 = 576 * 64
    const int per = (S + 3) >> 2, k0 = grp * per, k1 = k0 + per < S ? k0 + per : S;
    double s[4] = {0.0, 0.0, 0.0, 0.0};                    // the K-slices' fp32 sums are added up in float64, fixed order
    for (int k = k0; k < k1; ++k) {
        const f32x4 v = reinterpret_cast<const f32x4*>(partial + (size_t)k * 9 * WN_C * WN_C)[e4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += (double)v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[grp][lane][j] = s[j];
    __syncthreads();
    if (grp == 0) {
        f32x4 t;
        f32x4 wv = {0.f, 0.f, 0.f, 0.f};
        if (w) wv = reinterpret_cast<const f32x4*>(w)[e4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            t[j] = (float)(((red[0][lane][j] + red[1][lane][j]) + (red[2][lane][j] + red[3][lane][j])) + (double)wd * (double)wv[j]);
        reinterpret_cast<f32x4*>(dw)[e4] = t;
    }
}

static void ww_plan(int N, int H, int W, WwArgs* a, int* S) {
    a->N = N; a->H = H; a->W = W;
    a->tiles_x = ic_cdiv(W, 2); a->tiles_y = ic_cdiv(H, 2); a->chunks_x = ic_cdiv(a->tiles_x, WW_TC);
    const long long nch = (long long)N * a->tiles_y * a->chunks_x;
    a->nchunks = (int)nch;
    int s = nch < 64 ? (int)nch : 64;                     // 4 channel blocks x 64 slices = one work-group per CU
    a->chunks_per_slice = (int)((nch + s - 1) / s);
    *S = (int)((nch + a->chunks_per_slice - 1) / a->chunks_per_slice);
}

static bool ww_supported(int N, int H, int W) {
    return (W & 1) == 0 && (long long)N * WN_C * H * W * 4 < (1ll << 31) && (long long)N * ic_cdiv(H, 2) * ic_cdiv(W, 16) < (1ll << 30);
}

extern "C" size_t ic_conv3x3_c128_wgrad_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || !ww_supported(N, H, W)) return 0;
    WwArgs a{}; int S;
    ww_plan(N, H, W, &a, &S);
    return (size_t)S * 9 * WN_C * WN_C * sizeof(float);
}

extern "C" int ic_conv3x3_c128_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, const float* w, float wd,
                                         void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(x && dy && dw && workspace && N > 0 && H > 0 && W > 0);
    if (!ww_supported(N, H, W)) return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_conv3x3_c128_wgrad_workspace_bytes(N, H, W)) return IC_ERR_WORKSPACE;
    WwArgs a{}; int S;
    ww_plan(N, H, W, &a, &S);
    a.x = x; a.dy = dy; a.partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (WW_EIGHT_WAVES) hipLaunchKernelGGL(wino3x3_c128_wgrad8_kernel, dim3(4 * S), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(wino3x3_c128_wgrad_kernel, dim3(4 * S), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(9 * WN_C * WN_C / 256), dim3(256), 0, st, a.partial, S, w, wd, dw);   // 576 blocks x 64 float4
    IC_LAUNCH_CHECK();
    return IC_OK;
}
