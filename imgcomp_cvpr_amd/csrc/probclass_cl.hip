// Context model, inference pass for k = 24 feature maps with the intermediate activations CHANNELS-LAST in HBM
// (N, D, H, W, 24) instead of planar.  Same arithmetic as probclass.hip -- every output is the same fp32 chain over
// (8-channel group, tap, ascending channel) cut into the same four partial sums -- so logits and bit costs are bit-identical to
// the planar kernels' (tests compare them), to the block-wise pass and to the sequential decoder's.
//   reference: code/probclass.py:63-106 (bitcost), :185-196 (residual_block), :214-221 (_ResShallow._logits), :227-261 (conv3d)
//
// Why another layout (round 4; counters of the planar kernels in profiles/r03_counters.txt: matrix pipe 63-68 % busy, 0.93
// vector instructions beside each MFMA, one ds_read_b32 per MFMA, 48 + 48 scalar-width moves per thread to stage a brick):
//   * a voxel's 24 channels are 96 contiguous bytes: a brick is staged with 16-byte loads and 16-byte LDS writes (8.4 + 8.4 per
//     thread instead of 48 + 48), the permutation the MFMA operand order wants is a free renaming of registers on the way;
//   * the LDS brick is [kd][row][col][24] with every 8-channel group stored even channels first: lane (voxel j, k-half kh) reads
//     its B operands of FOUR k-steps (channels 8 c8 + 2 ks + kh) as ONE ds_read_b128 -- 42 LDS reads per 168 MFMAs, not 168;
//   * the epilogue holds 4 consecutive channels per lane and 8-channel group (rows 4 kh .. 4 kh + 3 of the accumulator): bias,
//     residual and output are 16-byte moves (3 + 3 + 3 per lane instead of 12 + 12 + 12).
// The planar kernels stay: training reads the planar feature volumes in its backward, k = 64 and pre-padded blocks use them.
#include "common.h"
#include "internal.h"

typedef float pcl_f32x16 __attribute__((ext_vector_type(16)));
typedef float pcl_f32x4 __attribute__((ext_vector_type(4)));
typedef float pcl_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pcl_u32x4 __attribute__((ext_vector_type(4)));

#define PCL_K 24
#define PCL_NT 14          // live taps of the "other" mask, order (kd,kh,kw) -- probclass.hip PC_NT
#define PCL_NP 4           // partial sums per output -- probclass.hip PC_NP
#ifndef PCL_NSTEP
#define PCL_NSTEP 2        // depth slices a work-group of a k -> k layer walks
#endif
__device__ __forceinline__ constexpr int pcl_tap_kd(int t) { return t < 9 ? 0 : 1; }
__device__ __forceinline__ constexpr int pcl_tap_kh(int t) { return t < 9 ? t / 3 : (t < 12 ? 0 : 1); }
__device__ __forceinline__ constexpr int pcl_tap_kw(int t) { return t < 9 ? t % 3 : (t < 12 ? t - 9 : t - 12); }

struct PclArgs {
    const float* in;         // conv0: q (N, qC, qh, qw) planar; else activations (N, D, H, W, 24)
    const float* w0;         // conv0: [2,3,3,1,24] TF layout
    const float* bias;
    const float* res;        // (N, RD, RH, RW, 24), read at (+2, +2, +2), or null
    float* out;              // (N, OD, OH, OW, 24); final layer: logits (N, OD, OH, OW, L) or null
    const int64_t* symbols;  // final layer with bits
    float* bits;
    int N, D, H, W, OD, OH, OW, RD, RH, RW, Cout, relu;
    int qC, qh, qw;
    float pad_value;
};

// ---- conv0: 1 -> 24, first mask (13 live taps), ReLU; one lane = one voxel, all 24 channels; probclass.hip pc_conv0_k24_kernel
// with the 24 results stored as the voxel's 96-byte run ----
__global__ __launch_bounds__(256) void pcl_conv0_kernel(const PclArgs a) {
    constexpr int K = PCL_K;
    const int n = blockIdx.z;
    const int ovol = a.OD * a.OH * a.OW;
    int v = blockIdx.x * 256 + threadIdx.x;
    const bool live = v < ovol;
    if (!live) v = ovol - 1;
    const int ox = v % a.OW, t = v / a.OW;
    const int oy = t % a.OH, od = t / a.OH;
    const int qhw = a.qh * a.qw;
    const int c0 = od - 4, y0 = oy - 4, x0 = ox - 4;
    bool okd[2], okh[3], okw[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) okd[i] = c0 + i >= 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { okh[i] = y0 + i >= 0 && y0 + i < a.qh; okw[i] = x0 + i >= 0 && x0 + i < a.qw; }
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * a.qC * qhw), 0, a.qC * qhw * 4, 0x00020000);
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
                const float* wp = a.w0 + ((kd * 3 + kh) * 3 + kw) * K;
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = fmaf(xv, wp[j], acc[j]);
            }
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * K * ovol), 0, K * ovol * 4, 0x00020000);
    const unsigned voff = live ? (unsigned)(v * K * 4) : 0x80000000u;
#pragma unroll
    for (int g = 0; g < K / 4; ++g) {
        pcl_f32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = acc[4 * g + i] + a.bias[4 * g + i];
            if (a.relu) x = fmaxf(x, 0.f);
            r[i] = x;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pcl_u32x4, r), yr, voff, 16 * g, 0);
    }
}

// stages the (2, TR + 2, TC + 2) x 24-channel brick whose corner is (od, y0, x0) into lds[pos][24]; PERM picks the order of every
// 8-channel group: 0 = (0,2,4,6 | 1,3,5,7) for the 32x32x2 layers, 1 = (0,4,1,5 | 2,6,3,7) for the 16x16x4 layer
template <int TR, int TC, int PERM>
__device__ __forceinline__ void pcl_stage_brick(float* lds, const float* xin_n, int vol_bytes, int od, int y0, int x0, int H, int W, int tid) {
    constexpr int S = TC + 2, DSP = (TR + 2) * S, NPOS = 2 * DSP, UNITS = NPOS * 3, NIT = (UNITS + 255) / 256;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xin_n, 0, vol_bytes, 0x00020000);
    pcl_f32x4 lo[NIT], hi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int u = tid + 256 * it;
        const int pp = u / 3, g = u - 3 * pp;
        const int kd = pp / DSP, pos = pp - kd * DSP;
        const int rr = pos / S, cc = pos - rr * S;
        const int iy = y0 + rr, ix = x0 + cc;
        const bool ok = u < UNITS && iy < H && ix < W;                       // od + kd < D always (VALID conv)
        const unsigned off = ok ? (unsigned)(((((od + kd) * H + iy) * W + ix) * PCL_K + 8 * g) * 4) : 0x80000000u;
        lo[it] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        hi[it] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 16, 0));
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int u = tid + 256 * it;
        if (u < UNITS) {
            pcl_f32x4 p0, p1;
            if (PERM == 0) { p0 = pcl_f32x4{lo[it][0], lo[it][2], hi[it][0], hi[it][2]}; p1 = pcl_f32x4{lo[it][1], lo[it][3], hi[it][1], hi[it][3]}; }
            else           { p0 = pcl_f32x4{lo[it][0], hi[it][0], lo[it][1], hi[it][1]}; p1 = pcl_f32x4{lo[it][2], hi[it][2], lo[it][3], hi[it][3]}; }
            *(pcl_f32x4*)&lds[8 * u] = p0;                                   // unit u = (position u / 3, group u % 3): 8 u = pos * 24 + 8 g
            *(pcl_f32x4*)&lds[8 * u + 4] = p1;
        }
    }
}

// one (TR + 2) x (TC + 2) x 24-channel plane (depth `plane`) into lds[pos][24], split into the request (loads into registers) and
// the LDS write so that a plane can be requested before a tile's MFMAs and written after them
template <int TR, int TC, int PERM>
struct PclPlane {
    static constexpr int S = TC + 2, DSP = (TR + 2) * S, UNITS = DSP * 3, NIT = (UNITS + 255) / 256;
    pcl_f32x4 lo[NIT], hi[NIT];
    __device__ __forceinline__ void request(const __amdgpu_buffer_rsrc_t xr, int plane, int y0, int x0, int H, int W, int tid) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it;
            const int pos = u / 3, g = u - 3 * pos;
            const int rr = pos / S, cc = pos - rr * S;
            const int iy = y0 + rr, ix = x0 + cc;
            const bool ok = u < UNITS && iy < H && ix < W;
            const unsigned off = ok ? (unsigned)((((plane * H + iy) * W + ix) * PCL_K + 8 * g) * 4) : 0x80000000u;
            lo[it] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
            hi[it] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 16, 0));
        }
    }
    __device__ __forceinline__ void write(float* slot, int tid) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + 256 * it;
            if (u < UNITS) {
                pcl_f32x4 p0, p1;
                if (PERM == 0) { p0 = pcl_f32x4{lo[it][0], lo[it][2], hi[it][0], hi[it][2]}; p1 = pcl_f32x4{lo[it][1], lo[it][3], hi[it][1], hi[it][3]}; }
                else           { p0 = pcl_f32x4{lo[it][0], hi[it][0], lo[it][1], hi[it][1]}; p1 = pcl_f32x4{lo[it][2], hi[it][2], lo[it][3], hi[it][3]}; }
                *(pcl_f32x4*)&slot[8 * u] = p0;
                *(pcl_f32x4*)&slot[8 * u + 4] = p1;
            }
        }
    }
};

// ---- k -> k layer ("other" mask) on v_mfma_f32_32x32x2_f32: probclass.hip pc_mfma_kernel<24, 24, 1, 4, TR, TC, false>.
// A work-group walks NSTEP consecutive depth slices of one (row, column) tile: slice d + 1 reuses the plane d + 1 that slice d
// staged, the one new plane is requested before slice d's MFMAs and written behind them (two-slot plane ring in LDS). ----
template <int TR, int TC, int NSTEP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pcl_mfma_kernel(const PclArgs a, const float* __restrict__ wpk) {
    constexpr int K = PCL_K, S = TC + 2, DSP = (TR + 2) * S, RD = 7, C8 = K / 8, NG = C8 * PCL_NT;
    static_assert(TR * TC <= 128 && TR * TC > 96, "one 32-voxel accumulator tile per wave (the last one may be ragged)");
    __shared__ __attribute__((aligned(16))) float lds[2 * DSP * K];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int tiles_x = (a.OW + TC - 1) / TC, tiles_y = (a.OH + TR - 1) / TR;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int od0 = (b / tiles_y) * NSTEP;
    const int nsteps = min(NSTEP, a.OD - od0);
    const int n = blockIdx.z;
    const int x0 = tx * TC, y0 = ty * TR;
    const int ivol = a.D * a.H * a.W;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * ivol * K), 0, ivol * K * 4, 0x00020000);

    const int j = lane & 31, kh = lane >> 5;
    const int q_raw = 32 * wn + j;
    const bool q_ok = TR * TC == 128 || q_raw < TR * TC;
    const int q = q_ok ? q_raw : 0;
    const int bbase = ((q / TC) * S + (q % TC)) * K + 4 * kh;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, NG * 1024, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    auto wload = [&](int gt) -> pcl_f32x4 {
        return __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wlane, gt * 1024, 0));
    };
    constexpr int STEPS = NG * 4;
    static_assert(STEPS % PCL_NP == 0, "K steps divide into the partial sums");
    pcl_f32x4 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wload(t);

    PclPlane<TR, TC, 0> pl;
    pl.request(xr, od0, y0, x0, a.H, a.W, tid);
    asm volatile("" ::: "memory");
    pl.write(lds, tid);
    pl.request(xr, od0 + 1, y0, x0, a.H, a.W, tid);
    asm volatile("" ::: "memory");
    pl.write(lds + DSP * K, tid);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int ovol = a.OD * a.OH * a.OW, rvol = a.RD * a.RH * a.RW;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * ovol * K), 0, ovol * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.res ? a.res + (size_t)n * rvol * K : a.bias), 0, a.res ? rvol * K * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, K * 4, 0x00020000);
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    const int oy = y0 + q / TC, ox = x0 + q % TC;
    const bool live = q_ok && oy < a.OH && ox < a.OW;

#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
        if (i >= nsteps) break;                                            // wave-uniform
        const int od = od0 + i;
        const bool next = i + 1 < nsteps;
        if (next) pl.request(xr, od + 2, y0, x0, a.H, a.W, tid);          // the one plane slice od + 1 lacks; lands under the MFMAs
        // plane od sits in slot i & 1, plane od + 1 in the other one
        auto tapoff_of = [i](int t) { return (((i + pcl_tap_kd(t)) & 1) * DSP + pcl_tap_kh(t) * S + pcl_tap_kw(t)) * K; };
        pcl_f32x16 accp[PCL_NP];
#pragma unroll
        for (int p = 0; p < PCL_NP; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) accp[p][r] = 0.f;
        pcl_f32x4 bq[2];
        bq[0] = *(const pcl_f32x4*)&lds[bbase + tapoff_of(0)];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c8 = g / PCL_NT, t = g % PCL_NT;
            {
                const int gn = g + RD - 2;                                 // filter fragments five (group, tap) steps ahead; behind the
                if (gn < NG) ring[gn % RD] = wload(gn);                    // last group they wrap to the next slice's first ones
                else if (next) ring[gn % RD] = wload(gn - NG);
            }
            if (g + 1 < NG) {
                const int c8n = (g + 1) / PCL_NT, tn1 = (g + 1) % PCL_NT;
                bq[(g + 1) & 1] = *(const pcl_f32x4*)&lds[bbase + 8 * c8n + tapoff_of(tn1)];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int step = g * 4 + ks, part = step / (STEPS / PCL_NP);
                accp[part] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[g % RD][ks], bq[g & 1][ks], accp[part], 0, 0, 0);
                if (step + 1 == 2 * (STEPS / PCL_NP)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[0][r] = accp[0][r] + accp[1][r];
                }
            }
            (void)c8; (void)t;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 // 1 MFMA
                if (m == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // the step's filter request
                if (m == 1 && g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // the next step's LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        pcl_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = accp[0][r] + (accp[2][r] + accp[3][r]);   // accp[0] already holds p0 + p1
        // epilogue: accumulator register 4 g + m = channel 8 g + 4 kh + m of voxel j: 16-byte bias / residual / output per group
        const unsigned ooff = live ? (unsigned)((((od * a.OH + oy) * a.OW + ox) * K + 4 * kh) * 4) : 0x80000000u;
        const unsigned roff = live ? (unsigned)(((((od + 2) * a.RH + oy + 2) * a.RW + ox + 2) * K + 4 * kh) * 4) : 0x80000000u;
        pcl_f32x4 bia[3], rsd[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bia[g] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(br, (unsigned)(16 * kh), 32 * g, 0));
            rsd[g] = __builtin_bit_cast(pcl_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 32 * g, 0));
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            pcl_f32x4 o;
#pragma unroll
            for (int m = 0; m < 4; ++m) o[m] = fmaxf(acc[4 * g + m] + bia[g][m], relu_lo) + rsd[g][m];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pcl_u32x4, o), yr, ooff, 32 * g, 0);
        }
        if (next) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // every wave has read plane od for the last time
            pl.write(lds + (i & 1) * DSP * K, tid);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---- final layer (k -> L <= 16 logits + cross-entropy) on v_mfma_f32_16x16x4_f32: probclass.hip pc_final16_kernel<24, 24> ----
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void pcl_final16_kernel(const PclArgs a, const float* __restrict__ wpk) {
    constexpr int K = PCL_K, TR = 8, TC = 16, S = TC + 2, DSP = (TR + 2) * S, NPOS = 2 * DSP, RD = 7, C8 = K / 8;
    __shared__ __attribute__((aligned(16))) float lds[NPOS * K];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.OW + TC - 1) / TC, tiles_y = (a.OH + TR - 1) / TR;
    int b = ic_xcd_run(blockIdx.x, gridDim.x);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; const int od = b / tiles_y;
    const int n = blockIdx.z;
    const int x0 = tx * TC, y0 = ty * TR;
    const int ivol = a.D * a.H * a.W;
    const float* __restrict__ xin = a.in + (size_t)n * ivol * K;
    const int nn = lane & 15, kq = lane >> 4;
    // B operand of this lane for N tile nt: position (2 wave + nt, nn), channels (8 c8 + kq, 8 c8 + 4 + kq) = one ds_read_b64
    const int bbase = ((2 * wave) * S + nn) * K + 2 * kq;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, C8 * PCL_NT * 512, 0x00020000);
    const unsigned wlane = (unsigned)lane * 8u;
    auto wload = [&](int gt) -> pcl_f32x2 {
        return __builtin_bit_cast(pcl_f32x2, __builtin_amdgcn_raw_buffer_load_b64(wr, wlane, gt * 512, 0));
    };
    constexpr int STEPS = C8 * PCL_NT * 2;
    static_assert(STEPS % PCL_NP == 0, "K steps divide into the partial sums");
    pcl_f32x4 accp[PCL_NP][2];
#pragma unroll
    for (int p = 0; p < PCL_NP; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) accp[p][t][r] = 0.f;
    pcl_f32x2 ring[RD];
#pragma unroll
    for (int t = 0; t < RD - 2; ++t) ring[t] = wload(t);

    pcl_stage_brick<TR, TC, 1>(lds, xin, ivol * K * 4, od, y0, x0, a.H, a.W, tid);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    auto tapoff_of = [](int t) { return (pcl_tap_kd(t) * DSP + pcl_tap_kh(t) * S + pcl_tap_kw(t)) * K; };
    pcl_f32x2 bq[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bq[0][nt] = *(const pcl_f32x2*)&lds[bbase + nt * S * K + tapoff_of(0)];
#pragma unroll
    for (int g = 0; g < C8 * PCL_NT; ++g) {
        const int c8 = g / PCL_NT, t = g % PCL_NT;
        const bool more = (c8 + 1) * 8 < K;
        {
            const int tn = t + RD - 2;
            if (tn < PCL_NT || more) ring[tn % RD] = wload(c8 * PCL_NT + tn);
        }
        if (g + 1 < C8 * PCL_NT) {
            const int c8n = (g + 1) / PCL_NT, tn1 = (g + 1) % PCL_NT;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bq[(g + 1) & 1][nt] = *(const pcl_f32x2*)&lds[bbase + 8 * c8n + nt * S * K + tapoff_of(tn1)];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int step = g * 2 + h, part = step / (STEPS / PCL_NP);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                accp[part][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[t % RD][h], bq[g & 1][nt][h], accp[part][nt], 0, 0, 0);
            if (step + 1 == 2 * (STEPS / PCL_NP)) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) accp[0][nt] = accp[0][nt] + accp[1][nt];
            }
        }
    }
    const float* __restrict__ biasp = a.bias;
    const int ovol = a.OD * a.OH * a.OW;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        pcl_f32x4 acc = accp[0][nt] + (accp[2][nt] + accp[3][nt]);
        float val[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * kq + r;
            val[r] = fmaxf(acc[r] + (co < a.Cout ? biasp[co] : 0.f), 0.f);    // final layer keeps conv3d's default ReLU
        }
        float lg[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int r = 0; r < 4; ++r) lg[4 * g4 + r] = __shfl(val[r], 16 * g4 + nn);
        const int oy = y0 + 2 * wave + nt, ox = x0 + nn;
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

static void pcl_launch_mfma(const PclArgs& a, const float* wpk, hipStream_t st) {
    // tile shape per layer: the one that covers the plane with the fewest work-groups (probclass.hip launch_pc_mfma)
    const int shapes[3][2] = {{8, 16}, {5, 25}, {6, 21}};
    int best = 0; long long best_cost = -1;
    for (int i = 0; i < 3; ++i) {
        const long long cost = (long long)ic_cdiv(a.OH, shapes[i][0]) * ic_cdiv(a.OW, shapes[i][1]);
        if (best_cost < 0 || cost < best_cost) { best = i; best_cost = cost; }
    }
    constexpr int NSTEP = PCL_NSTEP;
    dim3 g((unsigned)(ic_cdiv(a.OD, NSTEP) * best_cost), 1, a.N);
    if (best == 0) hipLaunchKernelGGL((pcl_mfma_kernel<8, 16, NSTEP>), g, dim3(256), 0, st, a, wpk);
    else if (best == 1) hipLaunchKernelGGL((pcl_mfma_kernel<5, 25, NSTEP>), g, dim3(256), 0, st, a, wpk);
    else hipLaunchKernelGGL((pcl_mfma_kernel<6, 21, NSTEP>), g, dim3(256), 0, st, a, wpk);
}

// byte offsets inside one image's feature volume are 31-bit
bool icx_pc_cl_supported(int k, int L, int C, int h, int w) {
    return k == PCL_K && L <= 16 && (long long)(C + 3) * (h + 6) * (w + 6) * PCL_K * 4 < (1ll << 31) && (long long)C * h * w * 4 < (1ll << 31);
}

// wt: the 8 TF-layout filter / bias pointers; pk1, pk2: 32-row fragments of the two k -> k layers, pk16: 16-row fragments of the
// final layer (ic_pc_pack_filters_f32); b0, b1, b2: the workspace's three feature volumes (same float counts as the planar pass)
int icx_pc_forward_cl(const float* q, const int64_t* symbols, const float* const* wt, const float* pk1, const float* pk2,
                      const float* pk16, int L, float pad_value, float* logits, float* bits, int N, int C, int h, int w,
                      float* b0, float* b1, float* b2, hipStream_t st) {
    PclArgs a{};
    a.N = N; a.pad_value = pad_value;
    a.in = q; a.w0 = wt[0]; a.bias = wt[1]; a.out = b0; a.Cout = PCL_K; a.relu = 1;
    a.D = C + 4; a.H = h + 8; a.W = w + 8; a.OD = C + 3; a.OH = h + 6; a.OW = w + 6; a.qC = C; a.qh = h; a.qw = w;
    hipLaunchKernelGGL(pcl_conv0_kernel, dim3((unsigned)ic_cdiv(a.OD * a.OH * a.OW, 256), 1, N), dim3(256), 0, st, a);
    // res1/conv1: ReLU
    a.in = b0; a.bias = wt[3]; a.out = b1; a.res = nullptr;
    a.D = C + 3; a.H = h + 6; a.W = w + 6; a.OD = C + 2; a.OH = h + 4; a.OW = w + 4; a.relu = 1;
    pcl_launch_mfma(a, pk1, st);
    // res1/conv2: linear, + conv0's output cropped [2:, 2:-2, 2:-2]
    a.in = b1; a.bias = wt[5]; a.out = b2; a.res = b0; a.RD = C + 3; a.RH = h + 6; a.RW = w + 6;
    a.D = C + 2; a.H = h + 4; a.W = w + 4; a.OD = C + 1; a.OH = h + 2; a.OW = w + 2; a.relu = 0;
    pcl_launch_mfma(a, pk2, st);
    // final: k -> L, ReLU, logits channels-last + bits
    a.in = b2; a.bias = wt[7]; a.out = logits; a.res = nullptr; a.symbols = symbols; a.bits = bits; a.Cout = L;
    a.D = C + 1; a.H = h + 2; a.W = w + 2; a.OD = C; a.OH = h; a.OW = w; a.relu = 1;
    hipLaunchKernelGGL(pcl_final16_kernel, dim3((unsigned)(a.OD * ic_cdiv(a.OH, 8) * ic_cdiv(a.OW, 16)), 1, N), dim3(256), 0, st, a, pk16);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
