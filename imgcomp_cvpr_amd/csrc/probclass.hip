// Context model ("probability classifier"), res_shallow architecture: four causally masked VALID
// (2,3,3) conv3d layers 1 -> k -> k -> k -> L over the (channel, row, col) symbol volume, one residual,
// logits for ALL positions in parallel, fused cross-entropy bit cost.
//   reference: code/probclass.py:63-106 (bitcost), :150-176 (masks), :185-196 (residual_block),
//              :214-221 (_ResShallow._logits), :227-261 (conv3d), :268-292 (pad_for_probclass3d)
//
// Geometry: the volume (N, D=C, h, w) is padded with pad_value by 4 in
// FRONT of D only and by 4 on each side of H and W (pad-on-load here, never materialised); each layer
// shrinks D by 1 and H, W by 2; the residual adds layer-0's output cropped [2:, 2:-2, 2:-2]; the final
// layer keeps conv3d's default ReLU (probclass.py:220,233).
// Masks: filter slice kd=0 is dense; slice kd=1 keeps the row above the centre and, in the centre row,
// the taps left of the centre (first layer) or left of and including the centre (other layers).  Dead
// taps are skipped, not multiplied by zero: 13 resp. 14 live taps of 18.
//
// One lane = one output voxel x COB output channels; per-output fp32 FMA chain in (ci, kd, kh, kw) order,
// identical for every position (no split reductions, no atomics) so that an incremental decoder can
// reproduce the same logits bit-for-bit later ("hard parts").
#include "common.h"

struct PcLayerArgs {
    const float* in;      // layer 0: q (N,C,h,w); else (N,Cin,D,H,W) planar
    const float* w;       // [2,3,3,Cin,Cout] TF layout, unmasked
    const float* bias;    // [Cout]
    const float* res;     // residual source (N,Cout,RD,RH,RW) planar, read at (+2,+2,+2), or null
    float* out;           // (N,Cout,OD,OH,OW) planar; final layer: logits (N,OD,OH,OW,Cout) or null
    const int64_t* symbols;  // final layer with bits: (N,OD,OH,OW)
    float* bits;             // final layer: (N,OD,OH,OW) or null
    int N, Cin, Cout, D, H, W, OD, OH, OW;   // D,H,W: (virtual, padded) input extent
    int RD, RH, RW;
    int qC, qh, qw;       // layer 0: un-padded extents
    float pad_value;
    int relu;
    int prepadded;        // layer 0: `in` already is the padded volume (N,D,H,W)
};

// CONTIG: Cout == COB (one channel block covers the layer): filter taps and bias are runs of COB consecutive scalars that
// the scalar unit fetches 8 at a time, instead of COB clamped single loads per tap
template <int COB, bool FIRST, bool FINAL, bool CONTIG = false>
__global__ __launch_bounds__(256) void pc_conv3d_kernel(const PcLayerArgs a) {
    const int n = blockIdx.z;
    const int co0 = blockIdx.y * COB;
    const int ovol = a.OD * a.OH * a.OW;
    int v = blockIdx.x * 256 + threadIdx.x;
    const bool live = v < ovol;
    if (!live) v = ovol - 1;
    const int ox = v % a.OW;
    const int t = v / a.OW;
    const int oy = t % a.OH, od = t / a.OH;

    float acc[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) acc[j] = 0.f;
    int wofs[COB];
#pragma unroll
    for (int j = 0; j < COB; ++j) wofs[j] = min(co0 + j, a.Cout - 1);

    const int HW = a.H * a.W;
    const int cstride = FIRST ? 0 : a.D * HW;
    const float* inn = FIRST ? a.in + (size_t)n * a.qC * a.qh * a.qw : a.in + (size_t)n * a.Cin * cstride;
    const int tapstride = a.Cin * a.Cout;

    for (int ci = 0; ci < a.Cin; ++ci) {
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // causal mask (probclass.py:150-176)
                    const bool dead = (kd == 1) && (kh == 2 || (kh == 1 && (FIRST ? kw >= 1 : kw >= 2)));
                    if (dead) continue;
                    float xv;
                    if (FIRST && a.prepadded) {
                        xv = a.in[(((size_t)n * a.D + od + kd) * a.H + oy + kh) * a.W + ox + kw];
                    } else if (FIRST) {
                        const int c = od + kd - 4, y = oy + kh - 4, x = ox + kw - 4;
                        const bool in = c >= 0 && y >= 0 && y < a.qh && x >= 0 && x < a.qw;   // c < qC always
                        xv = in ? inn[((size_t)c * a.qh + y) * a.qw + x] : a.pad_value;
                    } else {
                        xv = inn[(size_t)ci * cstride + (size_t)(od + kd) * HW + (oy + kh) * a.W + (ox + kw)];
                    }
                    const float* wp = a.w + (size_t)((kd * 3 + kh) * 3 + kw) * tapstride + (size_t)ci * a.Cout;
#pragma unroll
                    for (int j = 0; j < COB; ++j) acc[j] = fmaf(xv, CONTIG ? wp[j] : wp[wofs[j]], acc[j]);
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < COB; ++j) {
        float r = acc[j] + (CONTIG ? a.bias[j] : a.bias[wofs[j]]);
        if (a.relu) r = fmaxf(r, 0.f);
        if (a.res && co0 + j < a.Cout) {
            const size_t ro = (((size_t)n * a.Cout + co0 + j) * a.RD + od + 2) * a.RH * a.RW
                              + (size_t)(oy + 2) * a.RW + (ox + 2);
            r += a.res[ro];
        }
        acc[j] = r;
    }
    if (!FINAL) {
#pragma unroll
        for (int j = 0; j < COB; ++j)
            if (co0 + j < a.Cout) a.out[((size_t)n * a.Cout + co0 + j) * ovol + v] = acc[j];
    } else {
        // all Cout = L logits live in this lane (host guarantees gridDim.y == 1)
        const size_t vox = (size_t)n * ovol + v;
        if (a.out) {
#pragma unroll
            for (int j = 0; j < COB; ++j)
                if (j < a.Cout) a.out[vox * a.Cout + j] = acc[j];
        }
        if (a.bits) {
            // softmax_cross_entropy_with_logits(one_hot) * log2(e)   (probclass.py:100-104)
            float m = acc[0];
#pragma unroll
            for (int j = 1; j < COB; ++j) if (j < a.Cout) m = fmaxf(m, acc[j]);
            float s = 0.f, lsym = 0.f;
            const int sym = (int)a.symbols[vox];
#pragma unroll
            for (int j = 0; j < COB; ++j) {
                if (j < a.Cout) {
                    const float sh = acc[j] - m;
                    s += expf(sh);
                    if (j == sym) lsym = sh;
                }
            }
            a.bits[vox] = __fmul_rn(logf(s) - lsym, 1.44269504f);
        }
    }
}

// conv0 for k = 24 (1 -> 24 channels, first mask: 13 live taps) with the addressing of the generic kernel taken out of the
// tap loop.  One lane = one output voxel, all 24 channels: the same fmaf(x, w, acc) chain over the taps in (kd, kh, kw) order
// as pc_conv3d_kernel<24, true, ...> -- bit-identical.  What changes: the symbol volume is read through a buffer descriptor
// with ONE lane offset (the window's corner) and a scalar offset per tap; whether a tap lies in the pad region is the AND of
// three per-axis masks computed once per lane; the 24 stores take the channel as a scalar offset.  (The generic kernel spent
// ~15 vector instructions per tap on bounds checks and 64-bit addresses and 6 per store: 650 per voxel, now ~420.)
__global__ __launch_bounds__(256) void pc_conv0_k24_kernel(const PcLayerArgs a) {
    constexpr int K = 24;
    const int n = blockIdx.z;
    const int ovol = a.OD * a.OH * a.OW;
    int v = blockIdx.x * 256 + threadIdx.x;
    const bool live = v < ovol;
    if (!live) v = ovol - 1;
    const int ox = v % a.OW, t = v / a.OW;
    const int oy = t % a.OH, od = t / a.OH;
    const int qhw = a.qh * a.qw;
    // corner of the 2 x 3 x 3 window in the UNPADDED volume: (od - 4, oy - 4, ox - 4)
    const int c0 = od - 4, y0 = oy - 4, x0 = ox - 4;
    bool okd[2], okh[3], okw[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) okd[i] = c0 + i >= 0;                       // c0 + i < qC always
#pragma unroll
    for (int i = 0; i < 3; ++i) { okh[i] = y0 + i >= 0 && y0 + i < a.qh; okw[i] = x0 + i >= 0 && x0 + i < a.qw; }
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * a.qC * qhw), 0, a.qC * qhw * 4, 0x00020000);
    // the corner may lie before the volume (negative index); a tap inside the volume has a non-negative index, one in the
    // pad region gets the out-of-range offset (no access) and the pad value
    const int corner = (c0 * a.qh + y0) * a.qw + x0;
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.f;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                if (kd == 1 && (kh == 2 || (kh == 1 && kw >= 1))) continue;             // first mask (probclass.py:150-160)
                const bool in = okd[kd] && okh[kh] && okw[kw];
                const unsigned off = in ? (unsigned)((corner + (kd * a.qh + kh) * a.qw + kw) * 4) : 0x80000000u;
                const float ld = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qr, off, 0, 0));
                const float xv = in ? ld : a.pad_value;
                const float* wp = a.w + ((kd * 3 + kh) * 3 + kw) * K;                     // Cin = 1: [tap][co], wave-uniform
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = fmaf(xv, wp[j], acc[j]);
            }
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * K * ovol), 0, K * ovol * 4, 0x00020000);
    const unsigned voff = live ? (unsigned)(v * 4) : 0x80000000u;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float r = acc[j] + a.bias[j];
        if (a.relu) r = fmaxf(r, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, r), yr, voff, j * ovol * 4, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Matrix-core version of the k -> k and k -> L layers ("other" mask, 14 live taps).
// Same implicit-GEMM shape as the autoencoder convs: D[co][voxel] += A[co][kk] * B[kk][voxel] with
// kk = (8-channel chunk, tap, 2-channel k-step); fp32 MFMA = an fmaf chain in exactly that order, so the
// logits are still a fixed, position-independent fp32 expression (what an incremental decoder must match).
//   B: the (2, TR+2, TC+2) x KC-channel halo brick of the work-group's TR x TC voxels of ONE depth slice,
//      staged through LDS once per KC channels;
//   A: filter fragments packed per call into the workspace by pc_pack3_kernel (filters are tiny: 14 KB per
//      8-channel chunk), streamed through a 7-slot register ring.
// Output channels are padded to 32 per tile (24 -> 32, L = 6 -> 32): the padding costs matrix-pipe time
// (25 % / 81 %) but the layers are small; the VALU kernel above stays as the any-shape fallback.
// ------------------------------------------------------------------------------------------------
typedef float pc_f32x16 __attribute__((ext_vector_type(16)));
typedef float pc_f32x4 __attribute__((ext_vector_type(4)));

#define PC_NT 14          // live taps of the "other" mask, order (kd,kh,kw)
#define PC_NP 4           // partial sums per output (see pc_mfma_kernel)
__device__ __forceinline__ constexpr int pc_tap_kd(int t) { return t < 9 ? 0 : 1; }
__device__ __forceinline__ constexpr int pc_tap_kh(int t) { return t < 9 ? t / 3 : (t < 12 ? 0 : 1); }
__device__ __forceinline__ constexpr int pc_tap_kw(int t) { return t < 9 ? t % 3 : (t < 12 ? t - 9 : t - 12); }

// packed[((c8*14 + t)*NCOT + n)*256 + l*4 + j] = w[kd][kh][kw][ci = 8 c8 + 2j + (l>>5)][co = 32n + (l&31)] (0 if co >= Cout);
// the three matrix-core layers of one network in ONE launch (blockIdx.y = layer): three 5 us launches cost more than the
// packing itself
struct PcPack3 { const float* w[3]; float* out[3]; int cout[3], ncot[3], total[3]; };
__global__ void pc_pack3_kernel(const PcPack3 a, int Cin) {
    const int y = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total[y]) return;
    const int j = idx & 3, l = (idx >> 2) & 63;
    int r = idx >> 8;
    const int n = r % a.ncot[y]; r /= a.ncot[y];
    const int t = r % PC_NT, c8 = r / PC_NT;
    const int tap = (pc_tap_kd(t) * 3 + pc_tap_kh(t)) * 3 + pc_tap_kw(t);
    const int ci = 8 * c8 + 2 * j + (l >> 5), co = 32 * n + (l & 31);
    a.out[y][idx] = co < a.cout[y] ? a.w[y][((size_t)tap * Cin + ci) * a.cout[y] + co] : 0.f;
}

// the same fragments for the ADJOINT layer (data gradient): the GEMM's k axis runs over the forward layer's OUTPUT
// channels (padded with zeros up to KPAD), its rows over the forward layer's input channels:
//   packed[...] = w[tap][ci_fwd = 32 n + (l & 31)][co_fwd = 8 c8 + 2 j + (l >> 5)]
__global__ void pc_pack_adjoint_kernel(const float* __restrict__ w, float* __restrict__ out, int CinF, int CoutF, int NCOT,
                                       int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 3, l = (idx >> 2) & 63;
    int r = idx >> 8;
    const int n = r % NCOT; r /= NCOT;
    const int t = r % PC_NT, c8 = r / PC_NT;
    const int tap = (pc_tap_kd(t) * 3 + pc_tap_kh(t)) * 3 + pc_tap_kw(t);
    const int kq = 8 * c8 + 2 * j + (l >> 5), row = 32 * n + (l & 31);
    out[idx] = (row < CinF && kq < CoutF) ? w[((size_t)tap * CinF + row) * CoutF + kq] : 0.f;
}

// FLIP: the adjoint layer -- tap t reads the brick at the mirrored offset (1 - kd, 2 - kh, 2 - kw); with the input
// zero-padded by (1, 2, 2) on every side this is the data gradient of the forward layer (pc_bwd_data below).
template <int CIN, int KC, int WM, int WN, int TR, int TC, bool FINAL, bool FLIP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pc_mfma_kernel(const PcLayerArgs a, const float* __restrict__ wpk) {
    constexpr int S = TC + 2, DS = (TR + 2) * S, CS = 2 * DS, CHUNK = KC * CS;
    constexpr int NST = (CHUNK + 255) / 256;
    constexpr int NCH = CIN / KC, C8 = KC / 8, RD = 7;
    static_assert(WM * WN == 4 && TR * TC <= 32 * WN && TR * TC > 32 * (WN - 1), "one 32-voxel accumulator tile per wave (the last one may be ragged)");
    static_assert(CIN % KC == 0 && KC % 8 == 0, "channel chunking");
    static_assert(DS <= 256 && NST * 256 >= CHUNK + 64, "one brick plane per pass of the work-group, 64 spare floats behind the brick");
    __shared__ float lds[NST * 256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int tiles_x = (a.OW + TC - 1) / TC, tiles_y = (a.OH + TR - 1) / TR;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);             // contiguous runs of bricks per XCD: shared planes in one L2
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int od = b / tiles_y;
    const int n = blockIdx.z;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const size_t cstride = (size_t)a.D * HW;
    const float* __restrict__ xin = a.in + (size_t)n * CIN * cstride + (size_t)od * HW;

    // staging plan: the brick [KC][2][TR+2][TC+2] is loaded plane by plane ((ci, kd) = 2 KC planes of DS elements): lane
    // tid < DS owns position (tid / S, tid % S) of every plane, so a load is descriptor + one loop-invariant lane offset
    // + a scalar plane offset, and the LDS address is plane * DS + tid.  (The first version derived (ci, kd, row, col)
    // from a flat element index per load: 10 vector instructions per MFMA, mostly that index arithmetic.)  VALID conv:
    // positions past the volume get an out-of-range offset -> the buffer load returns 0.
    // one descriptor per KC-channel chunk (based at the chunk's first channel): byte offsets stay below 2^31 for any volume
    // whose KC-channel slab does -- the whole feature volume may be larger (res_shallow_64 on a 4K tile: 2.3 GB per layer)
    const int xr_bytes = (int)(((size_t)KC * cstride - (size_t)od * HW) * 4);
    unsigned loff;
    {
        const int rr = tid / S, cc = tid - rr * S;
        const int iy = y0 + rr, ix = x0 + cc;
        loff = (tid < DS && iy < a.H && ix < a.W) ? (unsigned)((iy * a.W + ix) * 4) : 0x80000000u;   // od + kd < D always
    }
    const int j = lane & 31, kh = lane >> 5;
    // tile shapes whose voxel count is not a multiple of 32 (5 x 25, 6 x 21: chosen per layer so that the tile grid wastes the
    // least of the plane) leave the last lanes of the last wave without a voxel: they compute voxel 0 again and store nothing
    const int q_raw = 32 * wn + j;
    const bool q_ok = TR * TC == 32 * WN || q_raw < TR * TC;
    const int q = q_ok ? q_raw : 0;
    const int boff = kh * CS + (q / TC) * S + (q % TC);
    const int cot = blockIdx.y * WM + wm, ncot = gridDim.y * WM;
    // filter fragments: float4 index ((group * PC_NT + tap) * ncot + cot) * 64 + lane -- one loop-invariant lane offset and
    // a SCALAR offset per (8-channel group, tap): no vector address arithmetic in the loop
    const int cot_u = __builtin_amdgcn_readfirstlane(cot);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (CIN / 8) * PC_NT * ncot * 1024, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    auto wload = [&](int gt) -> pc_f32x4 {                   // gt = group * PC_NT + tap
        return __builtin_bit_cast(pc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, (gt * ncot + cot_u) * 1024, 0));
    };

    // The K sequence (chunk, 8-channel group, tap, k-step) is cut into PC_NP contiguous parts with one accumulator each,
    // summed as (p0 + p1) + (p2 + p3) in the epilogue.  A dependent fp32 MFMA chain issues only every ~128 clocks; the
    // parts are what lets the sequential decoder (pc_dec_fused_kernel) run one output's sum on four waves at once and
    // still reproduce these logits bit for bit.
    constexpr int STEPS_PER_CHUNK = C8 * PC_NT * 4, STEPS = NCH * STEPS_PER_CHUNK;
    static_assert(STEPS % PC_NP == 0, "K steps divide into the partial sums");
    pc_f32x16 accp[PC_NP];
#pragma unroll
    for (int p = 0; p < PC_NP; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) accp[p][r] = 0.f;
    pc_f32x4 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wload(t);

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c > 0) __syncthreads();                       // everyone done reading the previous brick
        float stv[2 * KC];                                 // all loads first, then the writes under ONE predicate
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(xin + (size_t)c * KC * cstride), 0, xr_bytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < 2 * KC; ++p) {
            const int so = (int)((((size_t)(p / 2)) * cstride + (size_t)(p & 1) * HW) * 4);      // scalar: plane (ci, kd) of the chunk
            stv[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, loff, so, 0));
        }
        // ONE predicate around all the writes (a predicate per write made the compiler pair every load with its own wait +
        // branch; an unpredicated form with a per-lane stride cost a vector multiply-add per write -- 48 per chunk, and
        // vector instructions are paid in matrix-pipe time here): plane p of the brick goes to lds[p * DS + tid] through the
        // instruction's immediate offset
        asm volatile("" ::: "memory");
        if (tid < DS) {
#pragma unroll
            for (int p = 0; p < 2 * KC; ++p) lds[tid + p * DS] = stv[p];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // B operands of tap g + 1 are read while the MFMAs of tap g run (left alone the compiler reads each one right before
        // its use: load, wait, multiply); the first tap of a brick reads its own after the barrier
        auto tapoff_of = [](int t) {
            return FLIP ? (1 - pc_tap_kd(t)) * DS + (2 - pc_tap_kh(t)) * S + (2 - pc_tap_kw(t))
                        : pc_tap_kd(t) * DS + pc_tap_kh(t) * S + pc_tap_kw(t);
        };
        float bq[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[0][ks] = lds[boff + (2 * ks) * CS + tapoff_of(0)];
#pragma unroll
        for (int g = 0; g < C8 * PC_NT; ++g) {
            const int c8 = g / PC_NT, t = g % PC_NT;
            const bool more = (c * C8 + c8 + 1) * 8 < CIN;
            {
                const int tn = t + RD - 2;
                if (tn < PC_NT || more) ring[tn % RD] = wload((c * C8 + c8) * PC_NT + tn);
            }
            if (g + 1 < C8 * PC_NT) {
                const int c8n = (g + 1) / PC_NT, tn1 = (g + 1) % PC_NT;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bq[(g + 1) & 1][ks] = lds[boff + (8 * c8n + 2 * ks) * CS + tapoff_of(tn1)];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int step = c * STEPS_PER_CHUNK + g * 4 + ks, part = step / (STEPS / PC_NP);
                accp[part] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[t % RD][ks], bq[g & 1][ks], accp[part], 0, 0, 0);
                // fold finished parts as early as the summation order allows: fewer live accumulators
                if (step + 1 == 2 * (STEPS / PC_NP)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[0][r] = accp[0][r] + accp[1][r];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // 1 MFMA
                if (i == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // the tap's filter request
                if (g + 1 < C8 * PC_NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read of the next tap
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    pc_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = accp[0][r] + (accp[2][r] + accp[3][r]);       // accp[0] already holds p0 + p1

    // epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*kh output channel of tile `cot`, j = voxel
    const int oy = y0 + q / TC, ox = x0 + q % TC;
    const bool live = q_ok && oy < a.OH && ox < a.OW;
    const int ovol = a.OD * a.OH * a.OW;
    const int v = (od * a.OH + (live ? oy : 0)) * a.OW + (live ? ox : 0);
    float val[16];
    // Bias, residual and output go through buffer descriptors, one per group of 8 channels (registers 4g .. 4g+3 of both
    // half-waves): the channel is a SCALAR offset, a voxel outside the volume an out-of-range lane offset (loads return 0,
    // stores are dropped), a group past Cout a wave-uniform skip -- no per-channel address arithmetic, compares or exec
    // masks.  Every vector instruction here is paid in matrix-pipe time of the three other waves of the SIMD (PMC pass:
    // 2.7 vector instructions per MFMA in the first version of this epilogue, 56 % MFMA utilisation).
    const int rvol = a.RD * a.RH * a.RW;
    const unsigned ovoff = live ? (unsigned)((4 * kh * ovol + v) * 4) : 0x80000000u;
    const unsigned rvoff = live ? (unsigned)((4 * kh * rvol + (od + 2) * a.RH * a.RW + (oy + 2) * a.RW + ox + 2) * 4) : 0x80000000u;
    const int cot_s = __builtin_amdgcn_readfirstlane(cot);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * cot_s + 8 * g;                       // first channel of the group (wave-uniform)
        if (c0 >= a.Cout) {
#pragma unroll
            for (int i = 0; i < 4; ++i) val[4 * g + i] = 0.f;
            continue;
        }
        const int nch = a.Cout - c0 < 8 ? a.Cout - c0 : 8;
        const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias + c0), 0, nch * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.res ? a.res + ((size_t)n * a.Cout + c0) * rvol : a.bias), 0, a.res ? nch * rvol * 4 : 0, 0x00020000);
        float bia[4], rsd[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bia[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)(16 * kh), 4 * i, 0));
            rsd[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, rvoff, i * rvol * 4, 0));
        }
        const float relu_lo = a.relu ? 0.f : -__builtin_inff();
        if constexpr (!FINAL) {
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)n * a.Cout + c0) * ovol), 0, nch * ovol * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = fmaxf(acc[4 * g + i] + bia[i], relu_lo) + rsd[i];
                val[4 * g + i] = x;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), yr, ovoff, i * ovol * 4, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) val[4 * g + i] = fmaxf(acc[4 * g + i] + bia[i], relu_lo) + rsd[i];
        }
    }
    if (FINAL) {
        // logits of one voxel are split over the two half-waves (kh = 0: channels 0-3, 8-11, ...; kh = 1: 4-7, ...)
        float lg[16];                                  // channels 0..15 of this voxel (L <= 16)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mine = val[4 * g + i];                     // channel 8g + 4kh + i
                const float other = __shfl_xor(mine, 32);              // channel 8g + 4(1-kh) + i
                lg[8 * g + i] = kh == 0 ? mine : other;
                lg[8 * g + 4 + i] = kh == 0 ? other : mine;
            }
        if (kh == 0 && live) {
            const size_t vox = (size_t)n * ovol + v;
            if (a.out) {
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) if (c2 < a.Cout) a.out[vox * a.Cout + c2] = lg[c2];
            }
            if (a.bits) {
                float m = lg[0];
#pragma unroll
                for (int c2 = 1; c2 < 16; ++c2) if (c2 < a.Cout) m = fmaxf(m, lg[c2]);
                float ssum = 0.f, lsym = 0.f;
                const int sym = (int)a.symbols[vox];
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) {
                    if (c2 < a.Cout) {
                        const float shv = lg[c2] - m;
                        ssum += expf(shv);
                        if (c2 == sym) lsym = shv;
                    }
                }
                a.bits[vox] = __fmul_rn(logf(ssum) - lsym, 1.44269504f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Final layer (k -> L <= 16 logits + cross-entropy) on v_mfma_f32_16x16x4_f32: 16 output rows instead of 32 -- with L = 6
// the 32-row form spends 26 of its 32 rows on padding, this one 10 of 16, half the matrix-pipe time (40 -> ~22 us on a
// Kodak volume).  Same K order as pc_mfma_kernel (8-channel group, tap, ascending channel) and the same four partial
// sums (p0 + p1) + (p2 + p3): an fp32 MFMA is an fmaf chain in k order, so the logits are bit-identical to the 32-row
// form's (tests compare them) and to the sequential decoder's, which keeps the 32-row packing.
//   A: packed16[((c8 * 14 + t) * 64 + lane) * 2 + h] = w[tap t][ci = 8 c8 + 4 h + (lane >> 4)][co = lane & 15]
//   B: lane (voxel n = lane & 15 of a 16-voxel tile row, k = lane >> 4) reads channels 8 c8 + 4 h + k, h = 0, 1 of one
//      brick position as ONE ds_read_b64: LDS brick [c8][k][kd][position][h]
//   a work-group = 8 x 16 voxels of one depth slice, wave w = tile rows 2w, 2w + 1 (two N tiles sharing every A fragment).
// ------------------------------------------------------------------------------------------------
typedef float pc_f32x2 __attribute__((ext_vector_type(2)));

__global__ void pc_pack16_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int h = idx & 1, l = (idx >> 1) & 63;
    const int r = idx >> 7;
    const int t = r % PC_NT, c8 = r / PC_NT;
    const int tap = (pc_tap_kd(t) * 3 + pc_tap_kh(t)) * 3 + pc_tap_kw(t);
    const int ci = 8 * c8 + 4 * h + (l >> 4), co = l & 15;
    out[idx] = co < Cout ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
}

// LC: the number of logits when known at compile time (6 centres in every shipped configuration), 0 = a.Cout at run time.  Round 6: with
// LC = 6 the epilogue gathers 2 values per voxel instead of 16 and its softmax loops are straight-line code over 6 logits (the generic
// form runs 16-trip loops under run-time bounds on a quarter of the lanes: 3.0 vector instructions per MFMA, r03 counters) -- the same
// operations in the same order on the same values as LC = 0 (the sequential decoder's round-trip tests pin the logits bit for bit).
template <int CIN, int KC, int LC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pc_final16_kernel(const PcLayerArgs a, const float* __restrict__ wpk) {
    constexpr int TR = 8, TC = 16, S = TC + 2, DS = (TR + 2) * S, CHUNK = KC * 2 * DS;
    constexpr int NST = (CHUNK + 255) / 256, NCH = CIN / KC, C8 = KC / 8, RD = 7;
    static_assert(DS <= 256 && NST * 256 >= CHUNK + 64, "one brick plane per pass, 64 spare floats behind the brick");
    __shared__ __attribute__((aligned(16))) float lds[NST * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.OW + TC - 1) / TC, tiles_y = (a.OH + TR - 1) / TR;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int od = b / tiles_y;
    const int n = blockIdx.z;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const size_t cstride = (size_t)a.D * HW;
    const float* __restrict__ xin = a.in + (size_t)n * CIN * cstride + (size_t)od * HW;
    const int xr_bytes = (int)(((size_t)KC * cstride - (size_t)od * HW) * 4);
    unsigned loff;
    {
        const int rr = tid / S, cc = tid - rr * S;
        const int iy = y0 + rr, ix = x0 + cc;
        loff = (tid < DS && iy < a.H && ix < a.W) ? (unsigned)((iy * a.W + ix) * 4) : 0x80000000u;
    }
    const int nn = lane & 15, kq = lane >> 4;
    // B operand base of this lane for N tile nt: brick position (2 wave + nt, nn); float index of (c8 = 0, k = kq, kd = 0, pos, h = 0)
    const int bbase = (kq * 2 * DS + (2 * wave) * S + nn) * 2;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, (CIN / 8) * PC_NT * 512, 0x00020000);
    const unsigned wlane = (unsigned)lane * 8u;
    auto wload = [&](int gt) -> pc_f32x2 {                   // gt = group * PC_NT + tap: 512 bytes per (group, tap)
        return __builtin_bit_cast(pc_f32x2, __builtin_amdgcn_raw_buffer_load_b64(wr, wlane, gt * 512, 0));
    };
    constexpr int STEPS = (CIN / 8) * PC_NT * 2;             // 4-channel steps of the K sequence
    static_assert(STEPS % PC_NP == 0, "K steps divide into the partial sums");
    pc_f32x4 accp[PC_NP][2];
#pragma unroll
    for (int p = 0; p < PC_NP; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) accp[p][t][r] = 0.f;
    pc_f32x2 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wload(t);

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c > 0) __syncthreads();
        float stv[2 * KC];
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(xin + (size_t)c * KC * cstride), 0, xr_bytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < 2 * KC; ++p) {
            const int so = (int)((((size_t)(p / 2)) * cstride + (size_t)(p & 1) * HW) * 4);
            stv[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, loff, so, 0));
        }
        // brick [c8][k][kd][position][h]: plane (ci, kd) of the chunk goes to ((((ci >> 3) * 4 + (ci & 3)) * 2 + kd) * DS + pos) * 2 + ((ci >> 2) & 1);
        asm volatile("" ::: "memory");
        if (tid < DS) {                                        // one predicate around all the writes (see pc_mfma_kernel)
#pragma unroll
            for (int p = 0; p < 2 * KC; ++p) {
                const int ci = p >> 1, kd = p & 1;
                lds[2 * tid + ((((ci >> 3) * 4 + (ci & 3)) * 2 + kd) * DS) * 2 + ((ci >> 2) & 1)] = stv[p];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto tapoff_of = [](int t) { return (pc_tap_kd(t) * DS + pc_tap_kh(t) * S + pc_tap_kw(t)) * 2; };
        pc_f32x2 bq[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bq[0][nt] = *(const pc_f32x2*)&lds[bbase + nt * S * 2 + tapoff_of(0)];
#pragma unroll
        for (int g = 0; g < C8 * PC_NT; ++g) {
            const int c8 = g / PC_NT, t = g % PC_NT;
            const bool more = (c * C8 + c8 + 1) * 8 < CIN;
            {
                const int tn = t + RD - 2;
                if (tn < PC_NT || more) ring[tn % RD] = wload((c * C8 + c8) * PC_NT + tn);
            }
            if (g + 1 < C8 * PC_NT) {
                const int c8n = (g + 1) / PC_NT, tn1 = (g + 1) % PC_NT;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    bq[(g + 1) & 1][nt] = *(const pc_f32x2*)&lds[bbase + c8n * 4 * 2 * DS * 2 + nt * S * 2 + tapoff_of(tn1)];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int step = (c * C8 * PC_NT + g) * 2 + h, part = step / (STEPS / PC_NP);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    accp[part][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[t % RD][h], bq[g & 1][nt][h], accp[part][nt], 0, 0, 0);
                if (step + 1 == 2 * (STEPS / PC_NP)) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) accp[0][nt] = accp[0][nt] + accp[1][nt];
                }
            }
        }
    }
    // D rows 4 kq + r = output channel, column nn = voxel.  Channels 0..3 sit in the kq = 0 lanes, 4..7 in kq = 1, ...
    const float* __restrict__ biasp = a.bias;
    const int ovol = a.OD * a.OH * a.OW;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        pc_f32x4 acc = accp[0][nt] + (accp[2][nt] + accp[3][nt]);          // accp[0] already holds p0 + p1
        float val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * kq + r;
            val[r] = fmaxf(acc[r] + (co < a.Cout ? biasp[co] : 0.f), 0.f);    // final layer keeps conv3d's default ReLU
        }
        const int oy = y0 + 2 * wave + nt, ox = x0 + nn;
        if (LC == 6) {
            // channels 0..3 are this lane's own values in the kq = 0 lanes, 4 and 5 come from the kq = 1 lane of the same voxel
            const float l4 = __shfl(val[0], 16 + nn), l5 = __shfl(val[1], 16 + nn);
            if (kq == 0 && oy < a.OH && ox < a.OW) {
                const float lg6[6] = {val[0], val[1], val[2], val[3], l4, l5};
                const size_t vox = (size_t)n * ovol + ((size_t)od * a.OH + oy) * a.OW + ox;
                if (a.out) {
                    pc_f32x2* o2 = reinterpret_cast<pc_f32x2*>(a.out + vox * 6);           // 24 bytes per voxel: 8-byte aligned
                    o2[0] = pc_f32x2{lg6[0], lg6[1]}; o2[1] = pc_f32x2{lg6[2], lg6[3]}; o2[2] = pc_f32x2{lg6[4], lg6[5]};
                }
                if (a.bits) {
                    float m = lg6[0];
#pragma unroll
                    for (int c2 = 1; c2 < 6; ++c2) m = fmaxf(m, lg6[c2]);
                    float ssum = 0.f, lsym = 0.f;
                    const int sym = (int)a.symbols[vox];
#pragma unroll
                    for (int c2 = 0; c2 < 6; ++c2) {
                        const float shv = lg6[c2] - m;
                        ssum += expf(shv);
                        if (c2 == sym) lsym = shv;
                    }
                    a.bits[vox] = __fmul_rn(logf(ssum) - lsym, 1.44269504f);
                }
            }
            continue;
        }
        // gather the voxel's logits into its kq = 0 lane
        float lg[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[4 * g4 + r] = __shfl(val[r], 16 * g4 + nn);
        if (kq == 0 && oy < a.OH && ox < a.OW) {
            const size_t vox = (size_t)n * ovol + ((size_t)od * a.OH + oy) * a.OW + ox;
            if (a.out) {
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) if (c2 < a.Cout) a.out[vox * a.Cout + c2] = lg[c2];
            }
            if (a.bits) {
                float m = lg[0];
#pragma unroll
                for (int c2 = 1; c2 < 16; ++c2) if (c2 < a.Cout) m = fmaxf(m, lg[c2]);
                float ssum = 0.f, lsym = 0.f;
                const int sym = (int)a.symbols[vox];
#pragma unroll
                for (int c2 = 0; c2 < 16; ++c2) {
                    if (c2 < a.Cout) {
                        const float shv = lg[c2] - m;
                        ssum += expf(shv);
                        if (c2 == sym) lsym = shv;
                    }
                }
                a.bits[vox] = __fmul_rn(logf(ssum) - lsym, 1.44269504f);
            }
        }
    }
}

static size_t pc_packed16_floats(int k) { return (size_t)(k / 8) * PC_NT * 128; }
static size_t pc_packed_floats(int k, int cout) { return (size_t)(k / 8) * PC_NT * ic_cdiv(cout, 32) * 256; }
static bool pc_mfma_supported(int k, int L) { return (k == 24 || k == 64) && L <= 16; }

extern "C" size_t ic_pc_workspace_bytes(int N, int C, int h, int w, int k) {
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || k <= 0) return 0;
    size_t f = (size_t)(C + 3) * (h + 6) * (w + 6) + (size_t)(C + 2) * (h + 4) * (w + 4)
               + (size_t)(C + 1) * (h + 2) * (w + 2);
    size_t bytes = f * (size_t)N * k * sizeof(float);
    if (pc_mfma_supported(k, 16)) bytes += (2 * pc_packed_floats(k, k) + pc_packed_floats(k, 32) + pc_packed16_floats(k)) * sizeof(float);
    return bytes;
}

template <int COB, bool FIRST, bool FINAL, bool CONTIG = false>
static int launch_pc(const PcLayerArgs& a, hipStream_t st) {
    dim3 g(ic_cdiv(a.OD * a.OH * a.OW, 256), FINAL ? 1 : ic_cdiv(a.Cout, COB), a.N);
    hipLaunchKernelGGL((pc_conv3d_kernel<COB, FIRST, FINAL, CONTIG>), g, dim3(256), 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

static int launch_pc_mfma(const PcLayerArgs& a, const float* wpk, int k, bool final, hipStream_t st) {
    if (k == 24) {
        if (final) {
            dim3 g(a.OD * ic_cdiv(a.OH, 8) * ic_cdiv(a.OW, 16), 1, a.N);
            hipLaunchKernelGGL((pc_mfma_kernel<24, 24, 1, 4, 8, 16, true>), g, dim3(256), 0, st, a, wpk);
        } else {
            // tile shape per layer: a work-group computes TR x TC voxels and the launch runs ceil(OH / TR) x ceil(OW / TC) of them
            // per plane -- a 68 x 100 plane is 84 % useful in 8 x 16 tiles and 95 % in 5 x 25 (Kodak volume, layer 1)
            const int shapes[3][2] = {{8, 16}, {5, 25}, {6, 21}};
            int best = 0; long long best_cost = -1;
            for (int i = 0; i < 3; ++i) {
                const long long cost = (long long)ic_cdiv(a.OH, shapes[i][0]) * ic_cdiv(a.OW, shapes[i][1]);
                if (best_cost < 0 || cost < best_cost) { best = i; best_cost = cost; }
            }
            dim3 g((unsigned)(a.OD * best_cost), 1, a.N);
            if (best == 0) hipLaunchKernelGGL((pc_mfma_kernel<24, 24, 1, 4, 8, 16, false>), g, dim3(256), 0, st, a, wpk);
            else if (best == 1) hipLaunchKernelGGL((pc_mfma_kernel<24, 24, 1, 4, 5, 25, false>), g, dim3(256), 0, st, a, wpk);
            else hipLaunchKernelGGL((pc_mfma_kernel<24, 24, 1, 4, 6, 21, false>), g, dim3(256), 0, st, a, wpk);
        }
    } else {   // k == 64
        if (final) {
            dim3 g(a.OD * ic_cdiv(a.OH, 8) * ic_cdiv(a.OW, 16), 1, a.N);
            hipLaunchKernelGGL((pc_mfma_kernel<64, 16, 1, 4, 8, 16, true>), g, dim3(256), 0, st, a, wpk);
        } else {
            dim3 g(a.OD * ic_cdiv(a.OH, 4) * ic_cdiv(a.OW, 16), 1, a.N);
            hipLaunchKernelGGL((pc_mfma_kernel<64, 16, 2, 2, 4, 16, false>), g, dim3(256), 0, st, a, wpk);
        }
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// the three matrix-core filter packings [k->k | k->k | k->L] of one network, in one launch
static int pc_pack_filters(const float* const* wt, int k, int L, float* packed, hipStream_t st) {
    const int t1 = (int)pc_packed_floats(k, k), t3 = (int)pc_packed_floats(k, L);
    PcPack3 pa{};
    pa.w[0] = wt[2]; pa.w[1] = wt[4]; pa.w[2] = wt[6];
    pa.out[0] = packed; pa.out[1] = packed + t1; pa.out[2] = packed + 2 * (size_t)t1;
    pa.cout[0] = pa.cout[1] = k; pa.cout[2] = L;
    pa.ncot[0] = pa.ncot[1] = ic_cdiv(k, 32); pa.ncot[2] = 1;
    pa.total[0] = pa.total[1] = t1; pa.total[2] = t3;
    hipLaunchKernelGGL(pc_pack3_kernel, dim3(ic_cdiv(t1 > t3 ? t1 : t3, 256), 3), dim3(256), 0, st, pa, k);
    // the final layer once more in 16-row fragments (pc_final16_kernel); the 32-row packing stays for the sequential decoder
    const int t16 = (int)pc_packed16_floats(k);
    hipLaunchKernelGGL(pc_pack16_kernel, dim3(ic_cdiv(t16, 256)), dim3(256), 0, st, wt[6], packed + 2 * (size_t)t1 + pc_packed_floats(k, 32), k, L, t16);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// [k->k | k->k | k->L in 32-row fragments (room for 32 rows) | k->L in 16-row fragments]
extern "C" size_t ic_pc_packed_floats(int k, int L) {
    return pc_mfma_supported(k, L) ? 2 * pc_packed_floats(k, k) + pc_packed_floats(k, 32) + pc_packed16_floats(k) : 0;
}

extern "C" int ic_pc_pack_filters_f32(const float* const* wtab_host, int k, int L, float* packed, ic_stream_t stream) {
    IC_CHECK_ARG(wtab_host && packed && k > 0 && L > 0);
    for (int i = 0; i < 8; ++i) IC_CHECK_ARG(wtab_host[i] != nullptr);
    if (!pc_mfma_supported(k, L)) return IC_ERR_UNSUPPORTED;
    return pc_pack_filters(wtab_host, k, L, packed, (hipStream_t)stream);
}

// prepacked: the three MFMA filter packings already sit at the end of the workspace (the sequential decoder packs once
// per call).  wt[8], when not NULL, is a caller-owned packing made once at load time (ic_pc_pack_filters_f32): inference
// then runs without the per-call packing launch.
static int pc_forward(const float* q, int prepadded, const int64_t* symbols, const float* const* wt, int k, int L,
                      float pad_value, float* logits, float* bits, int N, int C, int h, int w,
                      void* workspace, size_t workspace_bytes, hipStream_t st, bool prepacked = false) {
    IC_CHECK_ARG(q && wt && workspace && N > 0 && C > 0 && h > 0 && w > 0 && k > 0 && L > 0);
    for (int i = 0; i < 8; ++i) IC_CHECK_ARG(wt[i] != nullptr);
    IC_CHECK_ARG(!bits || symbols);
    if (L > 16) return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_pc_workspace_bytes(N, C, h, w, k)) return IC_ERR_WORKSPACE;
    float* b0 = (float*)workspace;
    float* b1 = b0 + (size_t)N * k * (C + 3) * (h + 6) * (w + 6);
    float* b2 = b1 + (size_t)N * k * (C + 2) * (h + 4) * (w + 4);
    int rc;
    PcLayerArgs a{};
    a.N = N; a.pad_value = pad_value; a.prepadded = prepadded;
    // conv0: 1 -> k, first mask, ReLU
    a.in = q; a.w = wt[0]; a.bias = wt[1]; a.res = nullptr; a.out = b0;
    a.Cin = 1; a.Cout = k; a.D = C + 4; a.H = h + 8; a.W = w + 8; a.OD = C + 3; a.OH = h + 6; a.OW = w + 6;
    a.qC = C; a.qh = h; a.qw = w; a.relu = 1;
    // all k output channels of a voxel in one lane when k = 24: the input brick is read once instead of three times
    if (k == 24 && !prepadded && (long long)k * (C + 3) * (h + 6) * (w + 6) * 4 < (1ll << 31) && (long long)C * h * w * 4 < (1ll << 31)) {
        const int ovol0 = a.OD * a.OH * a.OW;
        hipLaunchKernelGGL(pc_conv0_k24_kernel, dim3((unsigned)((ovol0 + 255) / 256), 1, N), dim3(256), 0, st, a);
    } else if ((rc = (k == 24 ? launch_pc<24, true, false, true>(a, st) : launch_pc<8, true, false>(a, st)))) return rc;
    // (they address one KC-channel slab of an image's feature volume with 31-bit byte offsets; KC = 24 for k = 24, 16 for k = 64.
    // The same path must serve the parallel pass and the sequential decoder -- their logits have to agree bit for bit --
    // so the limit is the slab, not the volume: every volume a 288 GB device can hold stays on the matrix cores.)
    const bool use_mfma = pc_mfma_supported(k, L) && (size_t)(k == 24 ? 24 : 16) * (C + 3) * (h + 6) * (w + 6) * 4 < (1ull << 31);
    float* pk1 = wt[8] ? (float*)wt[8] : b2 + (size_t)N * k * (C + 1) * (h + 2) * (w + 2);
    float* pk2 = pk1 + pc_packed_floats(k, k);
    float* pk3 = pk2 + pc_packed_floats(k, k);
    if (use_mfma && !prepacked && !wt[8]) {
        if ((rc = pc_pack_filters(wt, k, L, pk1, st))) return rc;
    }
    // res1/conv1: k -> k, other mask, ReLU
    a.in = b0; a.w = wt[2]; a.bias = wt[3]; a.out = b1;
    a.Cin = k; a.D = C + 3; a.H = h + 6; a.W = w + 6; a.OD = C + 2; a.OH = h + 4; a.OW = w + 4; a.relu = 1;
    if ((rc = use_mfma ? launch_pc_mfma(a, pk1, k, false, st) : launch_pc<8, false, false>(a, st))) return rc;
    // res1/conv2: k -> k, linear, + conv0 output cropped [2:, 2:-2, 2:-2]
    a.in = b1; a.w = wt[4]; a.bias = wt[5]; a.out = b2; a.res = b0; a.RD = C + 3; a.RH = h + 6; a.RW = w + 6;
    a.D = C + 2; a.H = h + 4; a.W = w + 4; a.OD = C + 1; a.OH = h + 2; a.OW = w + 2; a.relu = 0;
    if ((rc = use_mfma ? launch_pc_mfma(a, pk2, k, false, st) : launch_pc<8, false, false>(a, st))) return rc;
    // conv2 (final): k -> L, ReLU (default activation), logits channels-last + bits
    a.in = b2; a.w = wt[6]; a.bias = wt[7]; a.out = logits; a.res = nullptr; a.symbols = symbols; a.bits = bits;
    a.Cout = L; a.D = C + 1; a.H = h + 2; a.W = w + 2; a.OD = C; a.OH = h; a.OW = w; a.relu = 1;
    if (use_mfma && !(k == 24 || k == 64)) rc = launch_pc_mfma(a, pk3, k, true, st);
    else if (use_mfma) {
        const float* pk16 = pk3 + pc_packed_floats(k, 32);
        dim3 g(a.OD * ic_cdiv(a.OH, 8) * ic_cdiv(a.OW, 16), 1, a.N);
#ifndef PC_FINAL_LC6
#define PC_FINAL_LC6 1          // 0: A/B builds with the run-time-L epilogue for L = 6 too
#endif
        if (k == 24 && L == 6 && PC_FINAL_LC6) hipLaunchKernelGGL((pc_final16_kernel<24, 24, 6>), g, dim3(256), 0, st, a, pk16);
        else if (k == 24) hipLaunchKernelGGL((pc_final16_kernel<24, 24>), g, dim3(256), 0, st, a, pk16);
        else if (L == 6 && PC_FINAL_LC6) hipLaunchKernelGGL((pc_final16_kernel<64, 16, 6>), g, dim3(256), 0, st, a, pk16);
        else hipLaunchKernelGGL((pc_final16_kernel<64, 16>), g, dim3(256), 0, st, a, pk16);
        IC_LAUNCH_CHECK();
        rc = IC_OK;
    }
    else if (L <= 8) rc = launch_pc<8, false, true>(a, st);
    else rc = launch_pc<16, false, true>(a, st);
    return rc;
}

extern "C" int ic_pc_logits_f32(const float* q, const float* const* wtab_host, int k, int L, float pad_value,
                                float* logits, int N, int C, int h, int w,
                                void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(logits);
    return pc_forward(q, 0, nullptr, wtab_host, k, L, pad_value, logits, nullptr, N, C, h, w,
                      workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ic_pc_logits_padded_f32(const float* vol, const float* const* wtab_host, int k, int L,
                                       float* logits, int N, int D, int H, int W,
                                       void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(logits);
    if (D < 5 || H < 9 || W < 9) return IC_ERR_UNSUPPORTED;
    return pc_forward(vol, 1, nullptr, wtab_host, k, L, 0.f, logits, nullptr, N, D - 4, H - 8, W - 8,
                      workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ic_pc_bitcost_f32(const float* q, const int64_t* symbols, const float* const* wtab_host, int k, int L,
                                 float pad_value, float* logits, float* bits, int N, int C, int h, int w,
                                 void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(bits && symbols);
    return pc_forward(q, 0, symbols, wtab_host, k, L, pad_value, logits, bits, N, C, h, w,
                      workspace, workspace_bytes, (hipStream_t)stream);
}

// ---- logits -> integer frequency tables for the arithmetic coder (probclass.py:443-444, :474) ----
// pr = softmax(logits); freqs = max(int64(pr * resolution), 1).  One lane per context; a fixed per-row fp32
// expression (max, exp, sequential sum, divide, multiply, truncate), so the encoder (all contexts at once) and
// the decoder (one context at a time) derive IDENTICAL tables from identical logits.
__device__ __forceinline__ void pc_table_row(const float* __restrict__ l, int L, float resolution, long long* __restrict__ freqs,
                                             float* __restrict__ pr) {
    float m = l[0];
    for (int j = 1; j < L; ++j) m = fmaxf(m, l[j]);
    float e[16];
    float s = 0.f;
    for (int j = 0; j < L; ++j) { e[j] = expf(l[j] - m); s += e[j]; }
    for (int j = 0; j < L; ++j) {
        const float p = e[j] / s;
        if (pr) pr[j] = p;
        long long f = (long long)__fmul_rn(p, resolution);
        freqs[j] = f < 1 ? 1 : f;
    }
}

__global__ __launch_bounds__(256) void logits_to_freqs_kernel(const float* __restrict__ logits, long long count, int L,
                                                              float resolution, long long* __restrict__ freqs,
                                                              float* __restrict__ pr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    pc_table_row(logits + i * L, L, resolution, freqs + i * L, pr ? pr + i * L : nullptr);
}

extern "C" int ic_pc_logits_to_freqs_f32(const float* logits, long long count, int L, float resolution,
                                         int64_t* freqs, float* pr, ic_stream_t stream) {
    IC_CHECK_ARG(logits && freqs && count > 0 && L > 0);
    if (L > 16) return IC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(logits_to_freqs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, count, L, resolution, (long long*)freqs, pr);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- data gradient of a k -> Cout_f layer on the matrix cores (training; train_pc.hip calls this) --------------------------
// dx[n][ci][u] = sum_{live taps t, co} g[n][co][u - off(t)] * w[t][ci][co]: a VALID conv of the (1,2,2)-zero-padded gradient
// with the mirrored taps and the transposed filter.  gpad: (N, KP, OD+2, OH+4, OW+4) with KP = 24 or 64 channels (channels
// >= Cout_f zero), dx_raw: (N, Cin_f, OD+1, OH+2, OW+2).  pk: pc_packed_floats(KP, Cin_f) floats of scratch.
__global__ __launch_bounds__(256) void pc_pad_grad_kernel(const float* __restrict__ g, float* __restrict__ gp, int N, int Cg, int KP,
                                                          int OD, int OH, int OW) {
    const int PD = OD + 2, PH = OH + 4, PW = OW + 4;
    const long long total = (long long)N * KP * PD * PH * PW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % PW);
    long long r = i / PW;
    const int y = (int)(r % PH); r /= PH;
    const int d = (int)(r % PD); r /= PD;
    const int c = (int)(r % KP), n = (int)(r / KP);
    const int od = d - 1, oy = y - 2, ox = x - 2;
    const bool in = c < Cg && od >= 0 && od < OD && oy >= 0 && oy < OH && ox >= 0 && ox < OW;
    gp[i] = in ? g[(((size_t)n * Cg + c) * OD + od) * OH * OW + (size_t)oy * OW + ox] : 0.f;
}

int icx_pc_bwd_data_mfma(const float* g, const float* w, float* dx_raw, int N, int CinF, int CoutF, int OD, int OH, int OW,
                         const float* zero_bias, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int KP = CoutF <= 24 ? 24 : 64;
    if (CoutF > 64 || CinF > 64 || (CinF != 24 && CinF != 64)) return IC_ERR_UNSUPPORTED;
    if ((size_t)KP * (OD + 2) * (OH + 4) * (OW + 4) * 4 >= (1ull << 31)) return IC_ERR_UNSUPPORTED;
    const size_t gp_floats = (size_t)N * KP * (OD + 2) * (OH + 4) * (OW + 4);
    const size_t pk_floats = pc_packed_floats(KP, CinF);
    if (workspace_bytes < (gp_floats + pk_floats) * sizeof(float)) return IC_ERR_WORKSPACE;
    float* gp = (float*)workspace;
    float* pk = gp + gp_floats;
    hipLaunchKernelGGL(pc_pad_grad_kernel, dim3((unsigned)((gp_floats + 255) / 256)), dim3(256), 0, st, g, gp, N, CoutF, KP,
                       OD, OH, OW);
    hipLaunchKernelGGL(pc_pack_adjoint_kernel, dim3(ic_cdiv((int)pk_floats, 256)), dim3(256), 0, st, w, pk, CinF, CoutF,
                       ic_cdiv(CinF, 32), (int)pk_floats);
    PcLayerArgs a{};
    a.in = gp; a.bias = zero_bias; a.res = nullptr; a.out = dx_raw;
    a.N = N; a.Cin = KP; a.Cout = CinF;
    a.D = OD + 2; a.H = OH + 4; a.W = OW + 4; a.OD = OD + 1; a.OH = OH + 2; a.OW = OW + 2; a.relu = 0;
    if (KP == 24) {
        dim3 grid(a.OD * ic_cdiv(a.OH, 8) * ic_cdiv(a.OW, 16), ic_cdiv(CinF, 32), N);
        hipLaunchKernelGGL((pc_mfma_kernel<24, 24, 1, 4, 8, 16, false, true>), grid, dim3(256), 0, st, a, pk);
    } else {
        dim3 grid(a.OD * ic_cdiv(a.OH, 4) * ic_cdiv(a.OW, 16), ic_cdiv(CinF, 64), N);
        hipLaunchKernelGGL((pc_mfma_kernel<64, 16, 2, 2, 4, 16, false, true>), grid, dim3(256), 0, st, a, pk);
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}

size_t icx_pc_bwd_data_mfma_workspace(int N, int CinF, int CoutF, int OD, int OH, int OW) {
    if (CoutF > 64 || (CinF != 24 && CinF != 64)) return 0;
    const int KP = CoutF <= 24 ? 24 : 64;
    if ((size_t)KP * (OD + 2) * (OH + 4) * (OW + 4) * 4 >= (1ull << 31)) return 0;
    return ((size_t)N * KP * (OD + 2) * (OH + 4) * (OW + 4) + pc_packed_floats(KP, CinF)) * sizeof(float);
}

// ---- sequential decoder (row N3: bit_counter.py:137-164 without the host in the loop) ----------------------------------
// A symbol's frequency table depends on the symbols decoded before it, so decoding is one context at a time by nature.
// The reference (and bit_counter._decode here) does a host round trip per symbol: gather the 5x9x9 context, run the
// network, fetch the table, step the arithmetic decoder in Python -- ~160 us per symbol.  Here the whole loop is
// enqueued on the stream: per symbol the SAME four context-model kernels as the parallel encoder side run on the
// gathered context (identical fp32 expression per logit -> identical tables, the property tested by
// test_blockwise_logits_bit_identical_to_full_volume), then ONE small kernel turns the logits into the integer table
// (pc_table_row, shared with logits_to_freqs_kernel), steps the arithmetic decoder (32-bit range coder, the reference's
// arithmetic_coding.py:ArithmeticDecoder restated for the device), stores the symbol, writes its centre into the padded
// volume and gathers the next context.  No host synchronisation until the end.
#define PC_AC_BITS 32
struct PcDecState {
    unsigned long long low, high, code;
    long long byte_pos;        // index of cur_byte (-1 before the first byte)
    int bit_left, cur_byte;
    int nxt_byte;              // byte byte_pos + 1, requested one byte early so that its load latency is off the path
    int error;                 // 1: frequency total too large, 2: internal
    long long next;            // raster index of the next symbol to decode
};

struct PcDecArgs {
    const unsigned char* bits; long long nbytes;
    PcDecState* st;
    const float* centers; const float* logits;
    float* vol;                // padded volume (C+4, h+8, w+8) of centre values
    float* ctx;                // (5, 9, 9) context of the next symbol
    long long* symbols;        // (C, h, w)
    int C, h, w, L, first_sym;
    float resolution;
};

__device__ __forceinline__ int pc_dec_bit(const PcDecArgs& a, PcDecState& s) {
    if (s.bit_left == 0) {
        s.cur_byte = s.nxt_byte;
        s.byte_pos += 1;
        const long long np = s.byte_pos + 1;
        s.nxt_byte = np < a.nbytes ? a.bits[np] : 0;   // past the end the stream reads as zeros (arithmetic_coding.py)
        s.bit_left = 8;
    }
    --s.bit_left;
    return (s.cur_byte >> s.bit_left) & 1;
}

__device__ void pc_dec_gather(const PcDecArgs& a, long long idx) {
    // context of symbol idx = padded block [c, c+5) x [y, y+9) x [x, x+9)
    const int HW = a.h * a.w;
    const int c = (int)(idx / HW), r = (int)(idx - (long long)c * HW);
    const int y = r / a.w, x = r - y * a.w;
    const int PH = a.h + 8, PW = a.w + 8;
    for (int e = threadIdx.x; e < 5 * 9 * 9; e += blockDim.x) {
        const int d = e / 81, r2 = e - d * 81;
        a.ctx[e] = a.vol[((size_t)(c + d) * PH + y + r2 / 9) * PW + x + r2 % 9];
    }
}

__global__ __launch_bounds__(256) void pc_dec_fill_kernel(float* __restrict__ vol, long long n, const float* __restrict__ centers) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) vol[i] = centers[0];                       // symbol 0 everywhere: pad_symbols_volume pads with 0
}

__global__ __launch_bounds__(256) void pc_dec_init_kernel(const PcDecArgs a) {
    if (threadIdx.x == 0) {
        PcDecState s;
        s.low = 0; s.high = (1ull << PC_AC_BITS) - 1; s.code = 0;
        s.byte_pos = -1; s.bit_left = 0; s.cur_byte = 0; s.nxt_byte = a.nbytes > 0 ? a.bits[0] : 0; s.error = 0; s.next = 1;
        for (int i = 0; i < PC_AC_BITS; ++i) s.code = (s.code << 1) | (unsigned)pc_dec_bit(a, s);
        *a.st = s;
        a.symbols[0] = a.first_sym;                       // the first symbol is not coded (bit_counter.py:117-121,152)
        a.vol[((size_t)4 * (a.h + 8) + 4) * (a.w + 8) + 4] = a.centers[a.first_sym];
    }
    __syncthreads();
    if ((long long)a.C * a.h * a.w > 1) pc_dec_gather(a, 1);
}

// one symbol: integer table from the logits, range-decoder step (arithmetic_coding.py:ArithmeticDecoder.read + _narrow)
__device__ int pc_dec_symbol(const PcDecArgs& a, PcDecState& s, const float* logits) {
    const unsigned long long MASK = (1ull << PC_AC_BITS) - 1, TOP = 1ull << (PC_AC_BITS - 1), SECOND = TOP >> 1;
    const unsigned long long MAX_TOTAL = (1ull << (PC_AC_BITS - 2)) + 2;
    long long fr[16];
    pc_table_row(logits, a.L, a.resolution, fr, nullptr);
    unsigned long long total = 0;
    for (int j = 0; j < a.L; ++j) total += (unsigned long long)fr[j];
    if (total > MAX_TOTAL) s.error = 1;
    const unsigned long long r = s.high - s.low + 1;
    const unsigned long long value = ((s.code - s.low + 1) * total - 1) / r;
    int sym = 0;
    unsigned long long cum = 0;
    while (sym + 1 < a.L && cum + (unsigned long long)fr[sym] <= value) { cum += (unsigned long long)fr[sym]; ++sym; }
    const unsigned long long cum_lo = cum, cum_hi = cum + (unsigned long long)fr[sym];
    s.high = s.low + cum_hi * r / total - 1;
    s.low = s.low + cum_lo * r / total;
    while (((s.low ^ s.high) & TOP) == 0) {
        s.code = ((s.code << 1) & MASK) | (unsigned)pc_dec_bit(a, s);
        s.low = (s.low << 1) & MASK;
        s.high = ((s.high << 1) & MASK) | 1;
    }
    while ((s.low & ~s.high & SECOND) != 0) {
        s.code = (s.code & TOP) | ((s.code << 1) & (MASK >> 1)) | (unsigned)pc_dec_bit(a, s);
        s.low = (s.low << 1) & (MASK >> 1);
        s.high = ((s.high << 1) & (MASK >> 1)) | TOP | 1;
    }
    return sym;
}

// n / d for n < 2^63, d < 2^34, n / d < 2^34: the double-precision quotient is within 2^-18 of the true one, so its integer
// part is off by at most one; one exact 64-bit multiply decides.  (The 64-bit integer division the compiler expands to is
// ~4x the instructions, and the sequential decoder does three per symbol on its critical path.)
__device__ __forceinline__ unsigned long long pc_udiv(unsigned long long n, unsigned long long d) {
    unsigned long long q = (unsigned long long)((double)n / (double)d);
    const long long rem = (long long)(n - q * d);
    if (rem < 0) --q; else if ((unsigned long long)rem >= d) ++q;
    return q;
}

// pc_dec_symbol for a whole wave: lane 48 + j holds logit j (0 beyond L -- logits are >= 0 after the ReLU, so the extra
// lanes do not move the maximum), the coder state is identical in every lane.  The per-symbol table is the expression of
// pc_table_row with its L exponentials, divisions and conversions spread over L lanes; the sum runs over readlane values in
// j order (0 + e0 + e1 + ...: the same fp32 sequence).  Returns the symbol (uniform).
template <int LC>       // LC = number of centres when known at compile time (the loops over readlane unroll), 0 = a.L
__device__ __forceinline__ int pc_dec_symbol_wave(const PcDecArgs& a, PcDecState& s, float logit) {
    const unsigned long long MASK = (1ull << PC_AC_BITS) - 1, TOP = 1ull << (PC_AC_BITS - 1), SECOND = TOP >> 1;
    const unsigned long long MAX_TOTAL = (1ull << (PC_AC_BITS - 2)) + 2;
    const int L = LC ? LC : a.L, lane = threadIdx.x & 63;
    auto bcast = [](float v, int src) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
    float m = bcast(logit, 48);
#pragma unroll
    for (int j = 1; j < L; ++j) m = fmaxf(m, bcast(logit, 48 + j));
    const float e = (lane >= 48 && lane < 48 + L) ? expf(logit - m) : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) sum += bcast(e, 48 + j);
    const float pr = e / sum;
    long long fl = (long long)__fmul_rn(pr, a.resolution);
    fl = fl < 1 ? 1 : fl;
    const unsigned f32 = (unsigned)fl;                   // <= resolution < 2^31 (total is checked against 2^30 + 2 below)
    unsigned long long total = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) total += (unsigned)__builtin_amdgcn_readlane((int)f32, 48 + j);
    if (total > MAX_TOTAL || (fl >> 31) != 0) s.error = 1;
    const unsigned long long r = s.high - s.low + 1;
    const unsigned long long value = pc_udiv((s.code - s.low + 1) * total - 1, r);
    int sym = 0;
    unsigned long long cum = 0;
    unsigned fs = (unsigned)__builtin_amdgcn_readlane((int)f32, 48);
    while (sym + 1 < L && cum + fs <= value) { cum += fs; ++sym; fs = (unsigned)__builtin_amdgcn_readlane((int)f32, 48 + sym); }
    const unsigned long long cum_lo = cum, cum_hi = cum + fs;
    s.high = s.low + pc_udiv(cum_hi * r, total) - 1;
    s.low = s.low + pc_udiv(cum_lo * r, total);
    while (((s.low ^ s.high) & TOP) == 0) {
        s.code = ((s.code << 1) & MASK) | (unsigned)pc_dec_bit(a, s);
        s.low = (s.low << 1) & MASK;
        s.high = ((s.high << 1) & MASK) | 1;
    }
    while ((s.low & ~s.high & SECOND) != 0) {
        s.code = (s.code & TOP) | ((s.code << 1) & (MASK >> 1)) | (unsigned)pc_dec_bit(a, s);
        s.low = (s.low << 1) & (MASK >> 1);
        s.high = ((s.high << 1) & (MASK >> 1)) | TOP | 1;
    }
    return sym;
}

__device__ __forceinline__ void pc_dec_store(const PcDecArgs& a, long long idx, int sym) {
    const int HW = a.h * a.w;
    const int c = (int)(idx / HW), rr = (int)(idx - (long long)c * HW);
    a.symbols[idx] = sym;
    a.vol[((size_t)(c + 4) * (a.h + 8) + rr / a.w + 4) * (a.w + 8) + rr % a.w + 4] = a.centers[sym];
}

__global__ __launch_bounds__(256) void pc_dec_step_kernel(const PcDecArgs a) {
    __shared__ long long sh_next;
    if (threadIdx.x == 0) {
        PcDecState s = *a.st;
        const int sym = pc_dec_symbol(a, s, a.logits);
        pc_dec_store(a, s.next, sym);
        s.next += 1;
        *a.st = s;
        sh_next = s.next;
        __threadfence_block();
    }
    __syncthreads();
    if (sh_next < (long long)a.C * a.h * a.w) pc_dec_gather(a, sh_next);
}

// ---- the same loop as ONE persistent work-group (k = 24): no launches between symbols -------------------------------------
// The five launches per symbol above cost ~5 us of launch latency each on top of ~5 us of work.  Here one work-group
// keeps the coder state in registers and the activations of the current 5x9x9 context in LDS and runs, per symbol:
// gather -> layer 0 (VALU, one lane per voxel) -> the three matrix-core layers -> table + range-decoder step.
// Bit-identical to the parallel pass by construction: every output is the same operation sequence -- layer 0 the fmaf
// chain of pc_conv3d_kernel<.., FIRST> in (kd,kh,kw) order, the other layers the MFMA chain of pc_mfma_kernel in
// (8-channel chunk, tap, k-step) order with the same packed A fragments, then + bias, ReLU, + residual.  Only the
// voxel -> lane assignment differs (the 75 / 18 / 1 output voxels of the context are packed densely into 32-voxel
// accumulator tiles: three waves, one wave, one wave), which no output value depends on.
struct PcFusedArgs {
    PcDecArgs d;
    const float* w0; const float* b0;          // layer 0: TF filter [2,3,3,1,k], bias
    const float* pk1; const float* b1;         // packed k -> k
    const float* pk2; const float* b2;
    const float* pk3; const float* b3;         // packed k -> L
    int* status;
};

// One of the PC_NP partial sums (steps [42 PART, 42 PART + 42) of the 168-step K sequence of pc_mfma_kernel<24, ...>) for
// NTL 32-voxel tiles at once: input volume [24][ID][IH][IW] in LDS, output voxel q = 32 i + (lane & 31) of the
// (ID-1, IH-2, IW-2) grid.  The tiles share the A fragments and give the wave independent accumulators to interleave.
template <int ID, int IH, int IW, int NTL, int PART>
__device__ __forceinline__ void pc_fused_part(const float* __restrict__ sin, const pc_f32x4* __restrict__ wp, int lane,
                                              pc_f32x16 (&acc)[NTL]) {
    constexpr int OH = IH - 2, OW = IW - 2, NV = (ID - 1) * OH * OW, IVOL = ID * IH * IW;
    constexpr int G0 = 42 * PART, G1 = G0 + 42, TS0 = G0 / 4, TS1 = (G1 - 1) / 4;       // tap-steps (c8 * 14 + t) touched
    static_assert(3 * PC_NT * 4 == 42 * PC_NP, "k = 24: 168 steps in 4 parts");
    const int kh = lane >> 5;
    int base[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int q = 32 * i + (lane & 31);
        const int qq = q < NV ? q : 0;
        const int od = qq / (OH * OW), oy = (qq / OW) % OH, ox = qq % OW;
        base[i] = (od * IH + oy) * IW + ox + kh * IVOL;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    pc_f32x4 av[TS1 - TS0 + 1];
#pragma unroll
    for (int ts = TS0; ts <= TS1; ++ts) av[ts - TS0] = wp[(size_t)ts * 64];               // all A fragments of the part up front
#pragma unroll
    for (int ts = TS0; ts <= TS1; ++ts) {
        const int c8 = ts / PC_NT, t = ts % PC_NT;
        const int tapoff = pc_tap_kd(t) * IH * IW + pc_tap_kh(t) * IW + pc_tap_kw(t);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int g = 4 * ts + ks;
            if (g < G0 || g >= G1) continue;
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
                const float bv = sin[base[i] + (8 * c8 + 2 * ks) * IVOL + tapoff];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ts - TS0][ks], bv, acc[i], 0, 0, 0);
            }
        }
    }
}

template <int ID, int IH, int IW, int NTL>
__device__ __forceinline__ void pc_fused_layer(const float* __restrict__ sin, const float* __restrict__ pk, int wave, int lane,
                                               pc_f32x16 (&acc)[NTL]) {
    const pc_f32x4* wp = reinterpret_cast<const pc_f32x4*>(pk) + lane;
    if (wave == 0) pc_fused_part<ID, IH, IW, NTL, 0>(sin, wp, lane, acc);
    else if (wave == 1) pc_fused_part<ID, IH, IW, NTL, 1>(sin, wp, lane, acc);
    else if (wave == 2) pc_fused_part<ID, IH, IW, NTL, 2>(sin, wp, lane, acc);
    else pc_fused_part<ID, IH, IW, NTL, 3>(sin, wp, lane, acc);
}

__global__ __launch_bounds__(256) void pc_dec_fused_kernel(const PcFusedArgs f) {
    constexpr int K = 24;
    __shared__ float s_ctx[5 * 9 * 9];
    __shared__ float s_a0[K * 196];            // layer 0 output  [k][4][7][7]
    __shared__ float s_a1[K * 75];             // res1/conv1      [k][3][5][5]
    __shared__ float s_a2[K * 18];             // res1/conv2 + skip [k][2][3][3]
    __shared__ float s_logits[16];
    __shared__ float s_red[4 * 16 * 64];       // the four partial accumulators of one tile, [part][register][lane]
    __shared__ float s_w0[13 * K];             // layer-0 filter rows of the 13 live taps, live-tap order
    __shared__ float s_bias[3 * K + 16];       // b0 | b1 | b2 | b3
    __shared__ float s_centers[16];
    const PcDecArgs& a = f.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 13 * K; e += 256) {
        const int lt = e / K;                                   // live tap lt -> (kd,kh,kw): 0..8 = kd 0; 9..11 = (1,0,*); 12 = (1,1,0)
        const int tap = lt < 9 ? lt : (lt < 12 ? 9 + (lt - 9) : 12);
        s_w0[e] = f.w0[(size_t)tap * K + e % K];
    }
    if (tid < K) { s_bias[tid] = f.b0[tid]; s_bias[K + tid] = f.b1[tid]; s_bias[2 * K + tid] = f.b2[tid]; }
    if (tid < a.L) { s_bias[3 * K + tid] = f.b3[tid]; s_centers[tid] = a.centers[tid]; }
    const long long n = (long long)a.C * a.h * a.w;
    PcDecState s;                               // lives in thread 0's registers for the whole volume
    if (tid == 0) {
        s.low = 0; s.high = (1ull << PC_AC_BITS) - 1; s.code = 0;
        s.byte_pos = -1; s.bit_left = 0; s.cur_byte = 0; s.nxt_byte = a.nbytes > 0 ? a.bits[0] : 0; s.error = 0; s.next = 1;
        for (int i = 0; i < PC_AC_BITS; ++i) s.code = (s.code << 1) | (unsigned)pc_dec_bit(a, s);
        pc_dec_store(a, 0, a.first_sym);        // the first symbol is not coded
    }
    __syncthreads();
    const int HW = a.h * a.w, PH = a.h + 8, PW = a.w + 8;
#ifdef PC_DEC_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define PC_PH(i) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tl; tl = tn_; } while (0)
#else
#define PC_PH(i) do { } while (0)
#endif
    for (long long idx = 1; idx < n; ++idx) {
        // ---- context of symbol idx: padded block [c, c+5) x [y, y+9) x [x, x+9) ----
        {
            const int c = (int)(idx / HW), r = (int)(idx - (long long)c * HW);
            const int y = r / a.w, x = r - y * a.w;
            for (int e = tid; e < 405; e += 256) {
                const int d = e / 81, r2 = e - d * 81;
                s_ctx[e] = a.vol[((size_t)(c + d) * PH + y + r2 / 9) * PW + x + r2 % 9];
            }
        }
        __syncthreads();
        PC_PH(0);
        // ---- layer 0: 1 -> k, first mask (13 live taps), + bias, ReLU; one lane = one of the 196 voxels ----
        if (tid < 196) {
            const int od = tid / 49, oy = (tid / 7) % 7, ox = tid % 7;
            float acc[K];
#pragma unroll
            for (int c = 0; c < K; ++c) acc[c] = 0.f;
#pragma unroll
            for (int kd = 0; kd < 2; ++kd)
#pragma unroll
                for (int kh2 = 0; kh2 < 3; ++kh2)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const bool dead = (kd == 1) && (kh2 == 2 || (kh2 == 1 && kw >= 1));
                        if (dead) continue;
                        const float xv = s_ctx[(od + kd) * 81 + (oy + kh2) * 9 + ox + kw];
                        const int tap = (kd * 3 + kh2) * 3 + kw;           // live taps are 0..12 in this order
                        const float* wp = s_w0 + tap * K;
#pragma unroll
                        for (int c = 0; c < K; ++c) acc[c] = fmaf(xv, wp[c], acc[c]);
                    }
#pragma unroll
            for (int c = 0; c < K; ++c) s_a0[c * 196 + tid] = fmaxf(acc[c] + s_bias[c], 0.f);
        }
        __syncthreads();
        PC_PH(1);
        // ---- res1/conv1: k -> k, ReLU; 75 voxels = 3 tiles; wave w computes partial sum w of all three ----
        {
            pc_f32x16 acc[3];
            pc_fused_layer<4, 7, 7, 3>(s_a0, f.pk1, wave, lane, acc);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s_red[(wave * 16 + r) * 64 + lane] = acc[i][r];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = tid + 256 * e, r = o >> 6, ln = o & 63;
                    const int co = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), q = 32 * i + (ln & 31);
                    const float v = (s_red[o] + s_red[1024 + o]) + (s_red[2048 + o] + s_red[3072 + o]);
                    if (co < K && q < 75) s_a1[co * 75 + q] = fmaxf(v + s_bias[K + co], 0.f);
                }
                __syncthreads();
            }
        }
        PC_PH(2);
        // ---- res1/conv2: k -> k, linear, + layer-0 output cropped [2:, 2:-2, 2:-2]; 18 voxels = 1 tile ----
        {
            pc_f32x16 acc[1];
            pc_fused_layer<3, 5, 5, 1>(s_a1, f.pk2, wave, lane, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[(wave * 16 + r) * 64 + lane] = acc[0][r];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = tid + 256 * e, r = o >> 6, ln = o & 63;
                const int co = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), q = ln & 31;
                const float v = (s_red[o] + s_red[1024 + o]) + (s_red[2048 + o] + s_red[3072 + o]);
                if (co < K && q < 18) {
                    const int od = q / 9, oy = (q / 3) % 3, ox = q % 3;
                    float x = v + s_bias[2 * K + co];
                    x += s_a0[co * 196 + ((od + 2) * 7 + oy + 2) * 7 + ox + 2];
                    s_a2[co * 18 + q] = x;
                }
            }
            __syncthreads();
        }
        PC_PH(3);
        // ---- conv2 (final): k -> L, ReLU; one voxel ----
        {
            pc_f32x16 acc[1];
            pc_fused_layer<2, 3, 3, 1>(s_a2, f.pk3, wave, lane, acc);
            if ((lane & 31) == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) s_red[(wave * 16 + r) * 64 + lane] = acc[0][r];
            }
            __syncthreads();
            if (tid < 16) {
                // channel co lives in register r = (co & 3) + 4 * (co >> 3) of lane 32 * ((co >> 2) & 1)
                const int co = tid, r = (co & 3) + 4 * (co >> 3), ln = 32 * ((co >> 2) & 1), o = r * 64 + ln;
                const float v = (s_red[o] + s_red[1024 + o]) + (s_red[2048 + o] + s_red[3072 + o]);
                if (co < a.L) s_logits[co] = fmaxf(v + s_bias[3 * K + co], 0.f);
            }
            __syncthreads();
        }
        PC_PH(4);
        if (tid == 0) {
            const int sym = pc_dec_symbol(a, s, s_logits);
            const int c = (int)(idx / HW), rr = (int)(idx - (long long)c * HW);
            a.symbols[idx] = sym;
            a.vol[((size_t)(c + 4) * PH + rr / a.w + 4) * PW + rr % a.w + 4] = s_centers[sym];
            __threadfence_block();
        }
        __syncthreads();
        PC_PH(5);
    }
    if (tid == 0) *f.status = s.error;
#ifdef PC_DEC_PROF
    if (tid == 0) for (int i = 0; i < 6; ++i) ((unsigned long long*)a.st)[i] = ph[i];
#endif
}

// ---- the persistent decoder with activation caches (k = 24): one NEW voxel per layer per symbol ----------------------------
// pc_dec_fused_kernel recomputes the whole 5x9x9 context of every symbol: 196 + 75 + 18 + 1 voxels.  But an activation depends
// only on symbols BEFORE its own position (the first layer's mask excludes the centre), so every voxel of every layer can
// be computed exactly once, the moment its last input is known, and kept (Fast-PixelCNN caching; here for a VALID-conv
// network over a padded volume, so the caches include the halo voxels that see pad values).  In absolute indices of the
// padded volume V[(C+4)][(h+8)][(w+8)]:
//     A0[d][i][j] = relu(conv0(V[d..d+1][i..i+2][j..j+2]))          d <= C+2, i <= h+5, j <= w+5
//     A1[d][i][j] = relu(conv1(A0[d..d+1][i..i+2][j..j+2]))         d <= C+1, i <= h+3, j <= w+3
//     A2[d][i][j] = conv2(A1[d..d+1][i..i+2][j..j+2]) + A0[d+2][i+2][j+2]
//     logits of symbol (c, y, x) = relu(conv3(A2[c..c+1][y..y+2][x..x+2]))
// and the last live tap of every window is its (1,1,1) corner ((1,1,0) for conv0).  The kernel sweeps P = (D, I, J) over the
// padded volume in raster order; at P it knows V[P] and computes  A0[D-1][I-1][J] -> A1[D-2][I-2][J-1] -> A2[D-3][I-3][J-2]
// -> the logits of the symbol at V[D][I][J+1], decodes it, and moves on.  Everything else those four need was computed at
// an earlier P: 13 of a window's 14 taps are prefetched from the caches (channels-last, in HBM/L2) while the range decoder
// works on the previous symbol, the (1,1,0) tap is the previous step's voxel and stays in LDS.
// Bit-identical to the parallel pass: fp32 MFMA is an fma chain in ascending k (tools/mfma_order.hip: v_mfma_f32_32x32x2_f32 and
// 16x16x4 against fmaf chains, 0 mismatches), so ONE output of pc_mfma_kernel is four fmaf chains over the K sequence
// (8-channel group, tap, channel pair) cut at 42-step boundaries, summed (p0 + p1) + (p2 + p3).  Wave w runs part w; lanes
// 0..23 hold the weights of conv1's outputs, lanes 24..47 conv2's, lanes 48.. conv3's -- the same 84 registers per lane serve
// all three layers, each layer is one pass of 84 dependent v_fma over broadcast LDS reads.
struct PcCachedArgs {
    PcDecArgs d;
    const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2; const float* w3; const float* b3;
    float* c0; float* c1; float* c2;          // activation caches, [d][i][j][24]
    int* status;
};

// chain position of (tap t, channel ci) in the K sequence of pc_mfma_kernel<24, ...>: 2 * (4 * ((ci / 8) * 14 + t) + (ci % 8) / 2) + ci % 2
__device__ __forceinline__ int pc_chain_idx(int t, int ci) { return 8 * ((ci >> 3) * PC_NT + t) + (ci & 7); }

__global__ __launch_bounds__(256) void pc_dec_cached_kernel(const PcCachedArgs f) {
    constexpr int K = 24, KT = PC_NT * K;                 // 336 inputs per output
    __shared__ __attribute__((aligned(16))) float s_in[3][KT];          // inputs of conv1 / conv2 / conv3 in chain order
    __shared__ __attribute__((aligned(16))) float s_v[16];              // the 13 live taps of conv0
    __shared__ float s_part[2][4][64];
    __shared__ float s_centers[16];
    const PcDecArgs& a = f.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = a.L;
    // ---- per-lane constants ----
    const int grp = lane < 24 ? 0 : (lane < 48 ? 1 : 2), co = lane - (grp == 2 ? 48 : 24 * grp), co0 = lane % 24;
    const int cout_g = grp == 2 ? L : K;
    const float* wl = grp == 0 ? f.w1 : (grp == 1 ? f.w2 : f.w3);
    float wreg[84];
#pragma unroll
    for (int n = 0; n < 84; ++n) {
        const int g = 42 * wave + (n >> 1), ts = g >> 2, c8 = ts / PC_NT, t = ts - c8 * PC_NT;
        const int ci = 8 * c8 + 2 * (g & 3) + (n & 1);
        const int tap = (pc_tap_kd(t) * 3 + pc_tap_kh(t)) * 3 + pc_tap_kw(t);
        wreg[n] = co < cout_g ? wl[((size_t)tap * K + ci) * cout_g + co] : 0.f;
    }
    float w0reg[13];
#pragma unroll
    for (int lt = 0; lt < 13; ++lt) w0reg[lt] = f.w0[lt * K + co0];     // live taps of the first mask are TF taps 0..12
    const float bias0 = f.b0[co0];
    const float bias_l = grp == 0 ? f.b1[co] : (grp == 1 ? f.b2[co] : (co < L ? f.b3[co] : 0.f));
    if (tid < L) s_centers[tid] = a.centers[tid];
    const float pad = a.centers[0];
    // ---- prefetch role: thread -> (layer pl, tap t < 12, channel quad q), and the conv0 taps on threads 0..11 ----
    const int pl = tid / 84, pe = tid - pl * 84, pt = pe / 6, pq = pe - pt * 6;
    const bool pf_on = tid < 252 && pt < 12;
    const int pkd = pt < 9 ? 0 : 1, pkh = pt < 9 ? pt / 3 : 0, pkw = pt < 9 ? pt % 3 : pt - 9;
    const int ni = a.h + 6 - 2 * pl, nj = a.w + 6 - 2 * pl;
    const float* cpl = pl == 0 ? f.c0 : (pl == 1 ? f.c1 : f.c2);
    const int pdst = 8 * ((pq >> 1) * PC_NT + pt) + 4 * (pq & 1);      // chain position of channels 4 pq .. 4 pq + 3 of tap pt
    const int vkd = tid < 9 ? 0 : 1, vkh = tid < 9 ? tid / 3 : 0, vkw = tid < 9 ? tid % 3 : tid - 9;
    const int PH = a.h + 8, PW = a.w + 8, HW = a.h * a.w;
    const int D1 = a.C + 3, I1 = a.h + 6, J1 = a.w + 5;   // last D, I, J of the sweep
    auto layer_valid = [&](int l, int D, int I, int J) -> bool {     // does step (D, I, J) produce a voxel of layer l + 1?
        return D >= 2 + l && I >= 2 + l && J >= 1 + l && I <= a.h + 5 - l && J <= a.w + 4 - l;
    };
    pc_f32x4 pf = {0.f, 0.f, 0.f, 0.f};
    float pv = pad;
    auto prefetch = [&](int D, int I, int J) {
        if (pf_on && layer_valid(pl, D, I, J)) {
            const int off = (((D - 2 - pl + pkd) * ni + (I - 2 - pl + pkh)) * nj + (J - 1 - pl + pkw)) * K + 4 * pq;
            pf = *reinterpret_cast<const pc_f32x4*>(cpl + off);
        }
        if (tid < 12) pv = a.vol[((size_t)(D - 1 + vkd) * PH + (I - 1 + vkh)) * PW + J + vkw];
    };
    PcDecState s;                                         // wave 0 keeps the coder state, identical in all its lanes
    s.low = 0; s.high = (1ull << PC_AC_BITS) - 1; s.code = 0;
    s.byte_pos = -1; s.bit_left = 0; s.cur_byte = 0; s.nxt_byte = a.nbytes > 0 ? a.bits[0] : 0; s.error = 0; s.next = 1;
    if (wave == 0)
        for (int i = 0; i < PC_AC_BITS; ++i) s.code = (s.code << 1) | (unsigned)pc_dec_bit(a, s);
    if (tid == 0) s_v[12] = pad;                          // V[1][1][0]
    // LDS hand-over between the waves: wait for this wave's LDS operations only.  (__syncthreads() also waits for the global
    // stores of the cache voxels to be acknowledged; their readers are a row of steps away and every wave drains its
    // memory counter at the top of each step, where it consumes its prefetch.)
    auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    int D = 1, I = 1, J = 0;
    prefetch(D, I, J);
    // one chain pass: this wave's part of the K sequence for the output this lane holds the weights of
    auto chain = [&](const float* in) -> float {
        const pc_f32x4* in4 = reinterpret_cast<const pc_f32x4*>(in + 84 * wave);
        float acc = 0.f;
#pragma unroll
        for (int n4 = 0; n4 < 21; ++n4) {
            const pc_f32x4 v = in4[n4];
            acc = fmaf(wreg[4 * n4], v[0], acc); acc = fmaf(wreg[4 * n4 + 1], v[1], acc);
            acc = fmaf(wreg[4 * n4 + 2], v[2], acc); acc = fmaf(wreg[4 * n4 + 3], v[3], acc);
        }
        return acc;
    };
#ifdef PC_DEC_PROF
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        PC_PH(6);
        // ---- the prefetched taps of this step -> LDS ----
        if (pf_on) *reinterpret_cast<pc_f32x4*>(&s_in[pl][pdst]) = pf;
        if (tid < 12) s_v[tid] = pv;
        lds_barrier();                                    // also: s_v[12] from the previous step's decode
        PC_PH(0);
        const bool v1 = layer_valid(0, D, I, J), v2 = layer_valid(1, D, I, J), v3 = layer_valid(2, D, I, J);
        // ---- conv0: every lane computes channel lane % 24 (so lane 24 + c holds the skip operand of conv2's output c) ----
        float a0 = 0.f;
        {
            const pc_f32x4* v4 = reinterpret_cast<const pc_f32x4*>(s_v);
            const pc_f32x4 va = v4[0], vb = v4[1], vc = v4[2], vd = v4[3];
            const float vv[13] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3], vc[0], vc[1], vc[2], vc[3], vd[0]};
#pragma unroll
            for (int lt = 0; lt < 13; ++lt) a0 = fmaf(vv[lt], w0reg[lt], a0);
        }
        a0 = fmaxf(a0 + bias0, 0.f);
        if (lane < K) {
            s_in[0][pc_chain_idx(13, lane)] = a0;         // every wave writes the same value: no barrier before its own reads
            if (wave == 0) f.c0[(((size_t)(D - 1) * (a.h + 6) + (I - 1)) * (a.w + 6) + J) * K + lane] = a0;
        }
        PC_PH(1);
        // ---- conv1 ----
        s_part[0][wave][lane] = chain(s_in[0]);
        lds_barrier();
        if (grp == 0) {
            const float v = (s_part[0][0][lane] + s_part[0][1][lane]) + (s_part[0][2][lane] + s_part[0][3][lane]);
            const float a1 = fmaxf(v + bias_l, 0.f);
            s_in[1][pc_chain_idx(13, co)] = a1;
            if (wave == 0 && v1) f.c1[(((size_t)(D - 2) * (a.h + 4) + (I - 2)) * (a.w + 4) + (J - 1)) * K + co] = a1;
        }
        PC_PH(2);
        // ---- conv2 + skip ----
        s_part[1][wave][lane] = chain(s_in[1]);
        lds_barrier();
        if (grp == 1) {
            const float v = (s_part[1][0][lane] + s_part[1][1][lane]) + (s_part[1][2][lane] + s_part[1][3][lane]);
            float a2 = v + bias_l;
            a2 += a0;
            s_in[2][pc_chain_idx(13, co)] = a2;
            if (wave == 0 && v2) f.c2[(((size_t)(D - 3) * (a.h + 2) + (I - 3)) * (a.w + 2) + (J - 2)) * K + co] = a2;
        }
        PC_PH(3);
        // ---- conv3 -> logits of the symbol at V[D][I][J + 1] ----
        float logit = 0.f;
        if (v3) {
            s_part[0][wave][lane] = chain(s_in[2]);
            lds_barrier();
            const float v = (s_part[0][0][lane] + s_part[0][1][lane]) + (s_part[0][2][lane] + s_part[0][3][lane]);
            logit = fmaxf(v + bias_l, 0.f);
        }
        PC_PH(4);
        // next step's coordinates
        int Dn = D, In = I, Jn = J + 1;
        if (Jn > J1) { Jn = 0; if (++In > I1) { In = 1; ++Dn; } }
        if (wave == 0) {
            float vnext = pad;
            if (v3) {
                const long long idx = ((long long)(D - 4) * a.h + (I - 4)) * a.w + (J - 3);
                const int sym = idx == 0 ? a.first_sym : (L == 6 ? pc_dec_symbol_wave<6>(a, s, logit) : pc_dec_symbol_wave<0>(a, s, logit));
                vnext = s_centers[sym];
                if (lane == 0) {
                    a.symbols[idx] = sym;
                    a.vol[((size_t)D * PH + I) * PW + J + 1] = vnext;
                }
            }
            if (lane == 0) s_v[12] = vnext;               // V at the next step's position (pad outside the symbol volume)
        }
        PC_PH(5);
        if (Dn > D1) break;
        // (1,1,0) taps of the next step = this step's voxels: centre slot -> tap-12 slot
        if (tid >= 64 && tid < 64 + 3 * K) {
            const int e = tid - 64, l2 = e / K, c = e - l2 * K;
            s_in[l2][pc_chain_idx(12, c)] = s_in[l2][pc_chain_idx(13, c)];
        }
        D = Dn; I = In; J = Jn;
        prefetch(D, I, J);
    }
    if (tid == 0) *f.status = s.error;
#ifdef PC_DEC_PROF
    if (tid == 0) printf("pc_dec_cached phases (clocks, thread 0): wait+stage %llu | conv0 %llu | conv1 %llu | conv2 %llu | conv3 %llu | decode %llu | tail+prefetch issue %llu\n",
                         ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
#endif
}

static size_t pc_dec_align(size_t b) { return (b + 255) & ~(size_t)255; }

// activation caches of pc_dec_cached_kernel (k = 24): the three feature volumes of the padded symbol volume, channels-last
static size_t pc_dec_cache_floats(int C, int h, int w, int k, int layer) {
    return (size_t)k * (C + 3 - layer) * (h + 6 - 2 * layer) * (w + 6 - 2 * layer);
}
static size_t pc_dec_cache_bytes(int C, int h, int w, int k) {
    if (k != 24) return 0;
    size_t b = 0;
    for (int l = 0; l < 3; ++l) b += pc_dec_align(pc_dec_cache_floats(C, h, w, k, l) * sizeof(float));
    return b;
}

extern "C" size_t ic_pc_decode_workspace_bytes(int C, int h, int w, int k) {
    if (C <= 0 || h <= 0 || w <= 0 || k <= 0) return 0;
    return pc_dec_align((size_t)(C + 4) * (h + 8) * (w + 8) * sizeof(float)) + pc_dec_align(405 * sizeof(float)) +
           pc_dec_align(16 * sizeof(float)) + pc_dec_align(sizeof(PcDecState)) + ic_pc_workspace_bytes(1, 1, 1, 1, k) +
           pc_dec_cache_bytes(C, h, w, k);
}

extern "C" int ic_pc_decode_f32(const uint8_t* bitstream, long long nbytes, int first_sym, const float* const* wtab_host,
                                const float* centers, int k, int L, float resolution, int64_t* symbols, int* status,
                                int C, int h, int w, void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(bitstream && wtab_host && centers && symbols && status && workspace);
    IC_CHECK_ARG(nbytes >= 0 && C > 0 && h > 0 && w > 0 && k > 0 && L > 0 && first_sym >= 0 && first_sym < L);
    if (L > 16) return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_pc_decode_workspace_bytes(C, h, w, k)) return IC_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)workspace;
    PcDecArgs a{};
    a.bits = bitstream; a.nbytes = nbytes; a.centers = centers; a.symbols = (long long*)symbols;
    a.C = C; a.h = h; a.w = w; a.L = L; a.first_sym = first_sym; a.resolution = resolution;
    const long long nvol = (long long)(C + 4) * (h + 8) * (w + 8);
    a.vol = (float*)p; p += pc_dec_align((size_t)nvol * sizeof(float));
    a.ctx = (float*)p; p += pc_dec_align(405 * sizeof(float));
    float* logits = (float*)p; p += pc_dec_align(16 * sizeof(float));
    a.logits = logits;
    a.st = (PcDecState*)p; p += pc_dec_align(sizeof(PcDecState));
    void* pcws = p;
    const size_t pcws_bytes = ic_pc_workspace_bytes(1, 1, 1, 1, k);
    p += pcws_bytes;
    hipLaunchKernelGGL(pc_dec_fill_kernel, dim3((unsigned)((nvol + 255) / 256)), dim3(256), 0, st, a.vol, nvol, centers);
    if (k == 24 && !(flags & (IC_PC_DECODE_PER_LAYER | IC_PC_DECODE_RECOMPUTE))) {
        PcCachedArgs f{};
        f.d = a;
        f.w0 = wtab_host[0]; f.b0 = wtab_host[1]; f.w1 = wtab_host[2]; f.b1 = wtab_host[3];
        f.w2 = wtab_host[4]; f.b2 = wtab_host[5]; f.w3 = wtab_host[6]; f.b3 = wtab_host[7];
        f.c0 = (float*)p; p += pc_dec_align(pc_dec_cache_floats(C, h, w, k, 0) * sizeof(float));
        f.c1 = (float*)p; p += pc_dec_align(pc_dec_cache_floats(C, h, w, k, 1) * sizeof(float));
        f.c2 = (float*)p;
        f.status = status;
        hipLaunchKernelGGL(pc_dec_cached_kernel, dim3(1), dim3(256), 0, st, f);
        IC_LAUNCH_CHECK();
        return IC_OK;
    }
    // filters packed once (pc_forward's own layout: after the three feature volumes of the 5x9x9 context)
    const bool use_mfma = pc_mfma_supported(k, L);
    if (use_mfma) {
        float* pk1 = (float*)pcws + (size_t)k * (4 * 7 * 7 + 3 * 5 * 5 + 2 * 3 * 3);
        const int rc = pc_pack_filters(wtab_host, k, L, pk1, st);
        if (rc) return rc;
    }
    if (use_mfma && k == 24 && !(flags & IC_PC_DECODE_PER_LAYER)) {
        PcFusedArgs f{};
        f.d = a;
        float* pk1 = (float*)pcws + (size_t)k * (4 * 7 * 7 + 3 * 5 * 5 + 2 * 3 * 3);
        f.w0 = wtab_host[0]; f.b0 = wtab_host[1];
        f.pk1 = pk1; f.b1 = wtab_host[3];
        f.pk2 = pk1 + pc_packed_floats(k, k); f.b2 = wtab_host[5];
        f.pk3 = pk1 + 2 * pc_packed_floats(k, k); f.b3 = wtab_host[7];
        f.status = status;
        hipLaunchKernelGGL(pc_dec_fused_kernel, dim3(1), dim3(256), 0, st, f);
        IC_LAUNCH_CHECK();
        return IC_OK;
    }
    hipLaunchKernelGGL(pc_dec_init_kernel, dim3(1), dim3(256), 0, st, a);
    IC_LAUNCH_CHECK();
    const long long n = (long long)C * h * w;
    auto one_symbol = [&]() -> int {
        int rc = pc_forward(a.ctx, 1, nullptr, wtab_host, k, L, 0.f, logits, nullptr, 1, 1, 1, 1, pcws, pcws_bytes, st, true);
        if (rc) return rc;
        hipLaunchKernelGGL(pc_dec_step_kernel, dim3(1), dim3(256), 0, st, a);
        return IC_OK;
    };
    // Every symbol runs the same five kernels with the SAME arguments (context, logits and coder state live at fixed
    // addresses), so a block of PC_DEC_GRAPH symbols is captured once into a hipGraph and replayed: the host cost of
    // ~1 M kernel launches per Kodak image (58 us per symbol, launch-bound) drops to one graph launch per block.
    long long i = 1;
    constexpr int PC_DEC_GRAPH = 128;
    if (n - 1 >= 2 * PC_DEC_GRAPH) {
        // capture is not allowed on the legacy default stream (torch's current stream by default): the loop runs on a
        // private stream ordered after / before the caller's by events
        // (created per call and destroyed before returning: the library keeps no per-process or per-device objects)
        hipStream_t own = nullptr;
        hipEvent_t ev_in = nullptr, ev_out = nullptr;
        bool ok = hipStreamCreateWithFlags(&own, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) == hipSuccess;
        auto release = [&]() {
            if (ev_in) (void)hipEventDestroy(ev_in);
            if (ev_out) (void)hipEventDestroy(ev_out);
            if (own) (void)hipStreamDestroy(own);
        };
        if (!ok) { release(); own = nullptr; (void)hipGetLastError(); }
        if (ok) {
            const hipStream_t caller = st;
            ok = hipEventRecord(ev_in, caller) == hipSuccess && hipStreamWaitEvent(own, ev_in, 0) == hipSuccess;
            if (ok) {
                st = own;
                int rc = IC_OK;
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
                    for (int j = 0; j < PC_DEC_GRAPH && rc == IC_OK; ++j) rc = one_symbol();
                    const hipError_t e = hipStreamEndCapture(st, &graph);
                    if (rc == IC_OK && e == hipSuccess && graph &&
                        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                        bool launched = true;
                        for (; launched && i + PC_DEC_GRAPH <= n; i += PC_DEC_GRAPH) launched = hipGraphLaunch(exec, st) == hipSuccess;
                        // the executable graph must outlive its last launch: wait before destroying it
                        (void)hipStreamSynchronize(st);
                        (void)hipGraphExecDestroy(exec);
                        if (!launched) rc = IC_ERR_ARG;
                    }
                    if (graph) (void)hipGraphDestroy(graph);
                }
                (void)hipGetLastError();
                for (; rc == IC_OK && i < n; ++i) rc = one_symbol();
                if (rc == IC_OK && hipMemcpyAsync(status, &a.st->error, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = IC_ERR_ARG;
                // the caller's stream continues after everything queued on the private one (also on the error paths)
                if (hipEventRecord(ev_out, own) != hipSuccess || hipStreamWaitEvent(caller, ev_out, 0) != hipSuccess) rc = rc ? rc : IC_ERR_ARG;
                if (rc == IC_OK) { const hipError_t e2 = hipGetLastError(); if (e2 != hipSuccess) rc = (int)e2; }
                release();           // destruction is deferred by the runtime until the queued work has drained
                return rc;
            }
            release();
        }
    }
    for (; i < n; ++i) {
        int rc = one_symbol();
        if (rc) return rc;
    }
    IC_LAUNCH_CHECK();
    if (hipMemcpyAsync(status, &a.st->error, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) return IC_ERR_ARG;
    return IC_OK;
}

// ---- deterministic sum (bits.py:4-14 numerator) ----
__global__ __launch_bounds__(256) void sum_stage1(const float* __restrict__ v, long long count, float* __restrict__ partial) {
    __shared__ float sh[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void sum_stage2(const float* __restrict__ partial, int n, float* __restrict__ out, float denom) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = denom != 0.f ? sh[0] / denom : sh[0];       // IEEE division: what `sum / n` computes in fp32
}

extern "C" int ic_sum_f32(const float* v, long long count, float* partial, float* out_sum, ic_stream_t stream) {
    IC_CHECK_ARG(v && partial && out_sum && count > 0);
    long long g = (count + 255) / 256;
    const int blocks = (int)(g > 1024 ? 1024 : g);
    hipLaunchKernelGGL(sum_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, count, partial);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, blocks, out_sum, 0.f);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// sum(v) / denom in the same two launches (bits.py:4-14: bpp = sum(bits) / num_pixels; the division rides in stage 2
// instead of a separate elementwise launch behind it)
extern "C" int ic_mean_f32(const float* v, long long count, float denom, float* partial, float* out, ic_stream_t stream) {
    IC_CHECK_ARG(v && partial && out && count > 0 && denom != 0.f);
    long long g = (count + 255) / 256;
    const int blocks = (int)(g > 1024 ? 1024 : g);
    hipLaunchKernelGGL(sum_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, count, partial);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, blocks, out, denom);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// out[r] = sum(v[r * count .. (r + 1) * count)) / denom for every row: ic_mean_f32 per row, one call across the ABI (val.py: the bpp of
// every image of a batch; `partial` is reused row after row on the one stream)
extern "C" int ic_mean_rows_f32(const float* v, int rows, long long count, float denom, float* partial, float* out, ic_stream_t stream) {
    IC_CHECK_ARG(v && partial && out && rows > 0 && count > 0);
    for (int r = 0; r < rows; ++r) {
        const int rc = ic_mean_f32(v + (size_t)r * count, count, denom, partial, out + r, stream);
        if (rc != IC_OK) return rc;
    }
    return IC_OK;
}

