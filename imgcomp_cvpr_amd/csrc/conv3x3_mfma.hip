// 3x3 / stride 1 / 128 -> 128 channel convolution + folded BN + ReLU + residual adds on the fp32 matrix
// cores of gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, one rounding per product, 64 FLOP/clk/SIMD).
//
// This is the dominant kernel of the path: the 64 residual convs of the CVPR autoencoder
// (reference code/autoencoder.py:274-287, called from :225-234 and :253-262) are ~95 % of its FLOPs.
//
// Formulation: implicit GEMM, transposed so that the MFMA's N axis is the pixel axis:
//     D[co][pixel] += A[co][k] * B[k][pixel],   k = (ci, tap)
// -> the accumulator's "col = lane & 31" axis is 32 neighbouring pixels, so every epilogue store /
//    residual load is a run of consecutive floats of one NCHW channel plane (coalesced), and the
//    BN scale/shift index (co) is constant per accumulator register.
//   A (filter): pre-packed once by ic_pack_conv3x3_c128_f32 into the exact per-lane fragment order;
//      each wave streams only ITS 32 output channels straight from L2 into registers with
//      global_load_dwordx4 (one load = the A operands of 4 MFMA k-steps); no LDS, no duplication.
//   B (activations): a (TR+2)x(TC+2) halo tile of 8 input channels at a time is staged through LDS
//      (register-staged ring); all 9 taps and all 4 waves re-read it from LDS with ds_read_b32.
// Work-group = 256 threads = 4 waves; wave w owns output channels [32w, 32w+32) for all TRxTC pixels
// of the tile (PT = TR*TC/32 accumulator tiles of 32x32 = 16*PT accumulator registers).
// K order per output: chunk (8 ci) -> tap (ky,kx) -> k-step (2 ci) ; fixed, independent of the tile
// position and of the tile variant, so results do not depend on the launch geometry.
//
// Two schedules of the same arithmetic (bit-identical results):
//   conv3x3_c128_kernel       simple double-buffered loop, compiler-scheduled (kept as the plain version)
//   conv3x3_c128_pipe_kernel  the production schedule, built from what the profiles showed:
//     * B operands of tap t+1 are fetched from LDS while the MFMAs of tap t issue (register double
//       buffer, one ds_read slotted behind every MFMA) -> one wave per SIMD keeps the matrix pipe fed;
//     * the filter fragments of chunk c+2 are requested the moment tap t of chunk c has consumed its
//       registers (two named register sets, two chunks = ~14k clocks of cover): with all 256 CUs
//       streaming the same 36 KB of filter per chunk in lock-step, an L2 load takes ~5k clocks;
//     * the LDS ring is 3 deep and its hand-over barrier sits in the MIDDLE of a chunk, so the last tap
//       of chunk c already prefetches tap 0 of chunk c+1 and the MFMA stream never drains;
//     * the epilogue first loads ALL its operands (BN scale/shift, residuals) through __restrict__
//       pointers, then computes and stores (a load->store->load chain costs one L2 round trip each).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define C128 128
#define KC 8                         // input channels per LDS chunk
#define NCHUNK (C128 / KC)           // 16
#define PACKED_FLOATS (NCHUNK * 9 * 4 * 64 * 4)

struct C3Args {
    const float* x; const float* wp; const float* scale; const float* shift;
    const float* res1; const float* res2; float* y;
    int N, H, W, tiles_x, tiles_y, relu;
    int abl;                   // tuning only: ablation mask (results invalid when != 0)
    unsigned long long* dbg;   // tuning only: 4 shader-clock stamps per work-group, or null
};

#define DBG_STAMP()                                                                               \
    do {                                                                                          \
        if (a.dbg && threadIdx.x == 0) {                                                          \
            unsigned long long* d_ = a.dbg + 4 * (size_t)blockIdx.x;                              \
            /* stamps: start, prologue length | (XCC_ID:HW_ID << 24), main-loop end, end */      \
            const unsigned long long hw_ = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | \
                                           (unsigned)__builtin_amdgcn_s_getreg(0xF804);            \
            d_[0] = t_start; d_[1] = (t_pro - t_start) | (hw_ << 24); d_[2] = t_main;             \
            d_[3] = __builtin_readcyclecounter();                                                 \
        }                                                                                         \
    } while (0)

// packed[(((c*9 + t)*4 + n)*64 + l)*4 + j] = w_tf[t][ci = 8c + 2j + (l>>5)][co = 32n + (l&31)]
__global__ void pack_conv3x3_c128_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    const int j = idx & 3, l = (idx >> 2) & 63, n = (idx >> 8) & 3;
    const int ct = idx >> 10;               // c*9 + t
    const int t = ct % 9, c = ct / 9;
    const int ci = KC * c + 2 * j + (l >> 5), co = 32 * n + (l & 31);
    out[idx] = w[(t * C128 + ci) * C128 + co];
}

// backward-data filter: dx = conv3x3(dy, W') with W'[t'][co][ci] = W[8 - t'][ci][co] (taps mirrored, channels swapped)
// packed[(((c*9 + t)*4 + n)*64 + l)*4 + j] = w_tf[8 - t][ci = 32n + (l&31)][co = 8c + 2j + (l>>5)]
__global__ void pack_conv3x3_c128_bwd_kernel(const float* __restrict__ w, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    const int j = idx & 3, l = (idx >> 2) & 63, n = (idx >> 8) & 3;
    const int ct = idx >> 10;
    const int t = ct % 9, c = ct / 9;
    const int k_in = KC * c + 2 * j + (l >> 5);      // reduction channel of the backward conv = forward OUTPUT channel
    const int m_out = 32 * n + (l & 31);             // produced channel = forward INPUT channel
    out[idx] = w[((8 - t) * C128 + m_out) * C128 + k_in];
}

// ------------------------------------------------------------------------------------------------
// shared pieces
// ------------------------------------------------------------------------------------------------
template <int TR, int TC>
struct Geo {
    static constexpr int S = TC + 2;               // LDS row stride (floats)
    static constexpr int CS = (TR + 2) * S;        // LDS channel stride
    static constexpr int CHUNK = KC * CS;          // floats per staged chunk
    static constexpr int NST = (CHUNK + 255) / 256;
    static constexpr int LDSF = NST * 256;         // padded: staging writes need no bounds branch
    static_assert(NST <= 32, "in-bounds mask is 32 bits");
};

// staging plan: element e = tid + 256 i of the chunk [ci][row][col] -> global offset + in-bounds bit
template <int TR, int TC>
__device__ __forceinline__ void staging_plan(int tid, int y0, int x0, int H, int W, int (&goff)[Geo<TR, TC>::NST],
                                             unsigned& inb) {
    using G = Geo<TR, TC>;
    inb = 0;
#pragma unroll
    for (int i = 0; i < G::NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / G::CS, rem = e - ci * G::CS;
        const int r = rem / G::S, c = rem - r * G::S;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = (e < G::CHUNK) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        goff[i] = ok ? ci * H * W + gy * W + gx : 0;
        inb |= (ok ? 1u : 0u) << i;
    }
}

// D[i][j]: i = (r&3) + 8*(r>>2) + 4*(lane>>5) is the output channel within the wave's 32, j = lane&31 the
// pixel within accumulator tile p.  All operand loads are issued before the first store.
template <int PT, int TC>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[PT], const C3Args& a, int n, int y0, int x0,
                                         int wave, int lane) {
    const float* __restrict__ scale = a.scale;
    const float* __restrict__ shift = a.shift;
    const float* __restrict__ res1 = a.res1;
    const float* __restrict__ res2 = a.res2;
    float* __restrict__ y = a.y;
    const int j = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
        sc[r] = scale[co];
        sh[r] = shift[co];
    }
    const size_t cbase = ((size_t)n * C128 + 32 * wave + 4 * kh) * HW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        const int oy = y0 + q / TC, ox = x0 + q % TC;
        const bool live = oy < a.H && ox < a.W;
        const size_t pix = cbase + (size_t)(live ? oy * a.W + ox : 0);
        float r1[16], r2[16];
        if (res1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) r1[r] = res1[pix + (size_t)((r & 3) + 8 * (r >> 2)) * HW];
        }
        if (res2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) r2[r] = res2[pix + (size_t)((r & 3) + 8 * (r >> 2)) * HW];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = fmaf(acc[p][r], sc[r], sh[r]);
            if (a.relu) v = fmaxf(v, 0.f);
            if (res1) v += r1[r];
            if (res2) v += r2[r];
            if (live) y[pix + (size_t)((r & 3) + 8 * (r >> 2)) * HW] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// plain schedule
// ------------------------------------------------------------------------------------------------
template <int PT, int TR, int TC>
__global__ __launch_bounds__(256) void conv3x3_c128_kernel(const C3Args a) {
    static_assert(TR * TC == 32 * PT, "tile must be PT MFMA pixel tiles");
    using G = Geo<TR, TC>;
    constexpr int S = G::S, CS = G::CS, NST = G::NST;
    __shared__ float lds[2][G::LDSF];

    const unsigned long long t_start = a.dbg ? __builtin_readcyclecounter() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; const int n = b / a.tiles_y;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * C128 * HW;

    int goff[NST];
    unsigned inb;
    staging_plan<TR, TC>(tid, y0, x0, a.H, a.W, goff, inb);

    const int j = lane & 31, kh = lane >> 5;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        boff[p] = kh * CS + (q / TC) * S + (q % TC);
    }
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + wave * 64 + lane;
    // wp[(c*9 + t)*256] = A fragments of chunk c, tap t for this wave's 32 output channels

    f32x16 acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x4 wcur[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wcur[t] = wp[t * 256];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const float v = xin[goff[i]];
        lds[0][tid + 256 * i] = ((inb >> i) & 1) ? v : 0.f;
    }
    __syncthreads();
    const unsigned long long t_pro = a.dbg ? __builtin_readcyclecounter() : 0ull;

    for (int c = 0; c < NCHUNK; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < NCHUNK;
        f32x4 wnext[9];
        float st[NST];
        if (more) {
            const f32x4* wn = wp + (size_t)(c + 1) * 9 * 256;
#pragma unroll
            for (int t = 0; t < 9; ++t) wnext[t] = wn[t * 256];
            const float* xc = xin + (size_t)(c + 1) * KC * HW;
#pragma unroll
            for (int i = 0; i < NST; ++i) { const float v = xc[goff[i]]; st[i] = ((inb >> i) & 1) ? v : 0.f; }
        }
        const float* __restrict__ L = lds[buf];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tapoff = (t / 3) * S + (t % 3);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float av = wcur[t][ks];
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const float bv = L[boff[p] + 2 * ks * CS + tapoff];
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p], 0, 0, 0);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NST; ++i) lds[buf ^ 1][tid + 256 * i] = st[i];
#pragma unroll
            for (int t = 0; t < 9; ++t) wcur[t] = wnext[t];
        }
        __syncthreads();
    }
    const unsigned long long t_main = a.dbg ? __builtin_readcyclecounter() : 0ull;
    epilogue<PT, TC>(acc, a, n, y0, x0, wave, lane);
    DBG_STAMP();
}

// ------------------------------------------------------------------------------------------------
// production schedule (see the header comment)
//   LDS ring WAR: buffer (c+1)%3 is written at the middle of chunk c; it was last read in chunk c-2 (and by
//   the tap-0 prefetch at the end of chunk c-3); every wave has since passed chunk c-1's barrier.
// ------------------------------------------------------------------------------------------------
template <int PT, int TR, int TC>
__global__ __launch_bounds__(256) void conv3x3_c128_pipe_kernel(const C3Args a) {
    static_assert(TR * TC == 32 * PT, "tile must be PT MFMA pixel tiles");
    using G = Geo<TR, TC>;
    constexpr int S = G::S, CS = G::CS, NST = G::NST;
    static_assert(NST <= 8, "halo-tile loads are spread over taps 5..8, two per tap");
    __shared__ float lds[3][G::LDSF];

    const unsigned long long t_start = a.dbg ? __builtin_readcyclecounter() : 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; const int n = b / a.tiles_y;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * C128 * HW;

    int goff[NST];
    unsigned inb;
    staging_plan<TR, TC>(tid, y0, x0, a.H, a.W, goff, inb);

    const int j = lane & 31, kh = lane >> 5;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        boff[p] = kh * CS + (q / TC) * S + (q % TC);
    }
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + ((a.abl & 4) ? 0 : wave * 64) + lane;

    f32x16 acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // filter fragments: set 0 holds even chunks, set 1 odd chunks
    f32x4 wq[2][9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[0][t] = wp[t * 256];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[1][t] = wp[(9 + t) * 256];
    float st[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const float v = xin[goff[i]];
        lds[0][tid + 256 * i] = ((inb >> i) & 1) ? v : 0.f;
    }
    {   // staging registers now carry chunk 1
        const float* xc = xin + (size_t)KC * HW;
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = xc[goff[i]];
    }
    __syncthreads();
    const unsigned long long t_pro = a.dbg ? __builtin_readcyclecounter() : 0ull;

    float bq[2][4][PT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < PT; ++p) bq[0][ks][p] = lds[0][boff[p] + 2 * ks * CS];

    int ring = 0;                                   // c % 3
    for (int c2 = 0; c2 < NCHUNK; c2 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {               // two chunks per trip: register sets stay statically named
            const int c = c2 + h;
            const bool more = c + 1 < NCHUNK;
            const bool more2 = c + 2 < NCHUNK && !(a.abl & 1);     // filter prefetch
            const bool more2s = c + 2 < NCHUNK && !(a.abl & 2);    // halo-tile prefetch
            const int ringn = ring == 2 ? 0 : ring + 1;
            const float* __restrict__ L = lds[ring];
            float* __restrict__ Ln = lds[ringn];
            const f32x4* wn2 = wp + (size_t)(c + 2) * 9 * 256;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cur = (h * 9 + t) & 1;
                if (t == 4) {
                    // hand chunk c+1 over through LDS, then start fetching chunk c+2's halo tile
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int i = 0; i < NST; ++i) Ln[tid + 256 * i] = ((inb >> i) & 1) ? st[i] : 0.f;
                    }
                    // raw barrier: __syncthreads() would also wait vmcnt(0) and drain the filter prefetch
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // Global loads are slotted one or two per tap into the MFMA stream (a VMEM issue that does not
                // fit into one MFMA's 64-clock shadow idles the matrix pipe of a one-wave-per-SIMD kernel):
                //  - halo tile of chunk c+2 into the staging registers freed by the hand-over above;
                //  - filter fragments of chunk c+2 into the registers of the tap TWO back (a load into registers
                //    that an MFMA issued < 64 clocks ago reads has to wait for that MFMA).
                if (t >= 5 && more2s) {
                    const float* xc = xin + (size_t)(c + 2) * KC * HW;
#pragma unroll
                    for (int i = 2 * (t - 5); i < 2 * (t - 5) + 2; ++i)
                        if (i < NST) st[i] = xc[goff[i]];        // raw; the zero-padding select is applied at the LDS write
                }
                if (t >= 2) {
                    if (more2) wq[h][t - 2] = wn2[(t - 2) * 256];
                } else if (c >= 1 && more && !(a.abl & 1)) {
                    wq[h ^ 1][7 + t] = wp[((size_t)(c + 1) * 9 + 7 + t) * 256];   // taps 7, 8 of chunk c-1 -> chunk c+1
                }
                if (t < 8) {
                    const int tapoff = ((t + 1) / 3) * S + ((t + 1) % 3);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int p = 0; p < PT; ++p) bq[cur ^ 1][ks][p] = L[boff[p] + 2 * ks * CS + tapoff];
                } else if (more) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int p = 0; p < PT; ++p) bq[cur ^ 1][ks][p] = Ln[boff[p] + 2 * ks * CS];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float av = wq[h][t][ks];
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[cur][ks][p], acc[p], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4 * PT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    if (i == PT || i == 2 * PT || i == 3 * PT)
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
                }
            }
            ring = ringn;
        }
    }
    const unsigned long long t_main = a.dbg ? __builtin_readcyclecounter() : 0ull;
    epilogue<PT, TC>(acc, a, n, y0, x0, wave, lane);
    DBG_STAMP();
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct C3Variant { int PT, TR, TC; };
static const C3Variant kVariants[] = {
    {4, 8, 16}, {4, 4, 32}, {3, 8, 12}, {3, 6, 16}, {3, 3, 32}, {2, 4, 16}, {2, 2, 32}, {2, 8, 8}, {1, 2, 16}, {1, 4, 8},
};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
// No process-wide state: the tile variant of a launch comes from the per-call flags (IC_CONV3_DIRECT_VARIANT(v), tests);
// the schedule knobs of the tuning tools exist only in -DIC_TUNING builds.
#ifdef IC_TUNING
static int g_lds_pad = 0;      // extra dynamic LDS per work-group (limits work-groups per CU)
static unsigned long long* g_dbg = nullptr;
static int g_pipe = -1;        // -1: automatic (pipelined schedule for PT >= 2, plain for PT == 1), 0: plain, 1: pipelined
static int g_abl = 0;
extern "C" void ic_conv3x3_c128_debug_set_buffer(void* p) { g_dbg = (unsigned long long*)p; }
extern "C" int ic_conv3x3_c128_debug_set_tuning(int key, int value) {
    int prev = -1;
    switch (key) {
        case 1: prev = g_lds_pad; g_lds_pad = value < 0 ? 0 : value; break;
        case 2: prev = g_pipe; g_pipe = value < 0 ? -1 : (value ? 1 : 0); break;
        case 3: prev = g_abl; g_abl = value; break;
        default: break;
    }
    return prev;
}
#else
static constexpr int g_lds_pad = 0, g_pipe = -1, g_abl = 0;
static constexpr unsigned long long* g_dbg = nullptr;
#endif

// Variant choice from a small cost model calibrated on MI355X (tools/kbench.py conv3 with IC_CONV3_DIRECT_VARIANT(v), profiles/):
// every CU gets W = ceil(nwg / 256) work-groups; up to `occ` of them are resident together, and the
// matrix-pipe efficiency of a CU grows with the number of resident waves per SIMD (1: 0.70 with the
// pipelined schedule, 2: 0.85, >= 3: 0.95 -- prologue/epilogue of one group hide under the others).
static int variant_occupancy(int PT) { return PT == 1 ? 3 : (PT <= 3 ? 2 : 1); }

static int pick_variant(int N, int H, int W, int flags) {
    const int forced = ((flags >> 8) & 0xf) - 1;                  // IC_CONV3_DIRECT_VARIANT(v)
    if (forced >= 0 && forced < kNumVariants) return forced;
    static const double eff[4] = {0.0, 0.70, 0.85, 0.95};
    int best = 0; double bestc = 1e30;
    for (int v = 0; v < kNumVariants; ++v) {
        const C3Variant& k = kVariants[v];
        const long nwg = (long)N * ic_cdiv(H, k.TR) * ic_cdiv(W, k.TC);
        const long per_cu = (nwg + 255) / 256;
        const int conc = (int)(per_cu < variant_occupancy(k.PT) ? per_cu : variant_occupancy(k.PT));
        const double cost = (double)per_cu * (k.PT + 0.3) / eff[conc];
        if (cost < bestc - 1e-9) { bestc = cost; best = v; }
    }
    return best;
}

extern "C" size_t ic_conv3x3_c128_packed_floats(void) { return PACKED_FLOATS; }

extern "C" int ic_pack_conv3x3_c128_bwd_f32(const float* w_tf, float* w_packed, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    hipLaunchKernelGGL(pack_conv3x3_c128_bwd_kernel, dim3(PACKED_FLOATS / 256), dim3(256), 0, (hipStream_t)stream,
                       w_tf, w_packed);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

extern "C" int ic_pack_conv3x3_c128_f32(const float* w_tf, float* w_packed, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    hipLaunchKernelGGL(pack_conv3x3_c128_kernel, dim3(PACKED_FLOATS / 256), dim3(256), 0, (hipStream_t)stream,
                       w_tf, w_packed);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

#define C3_LAUNCH(PT_, TR_, TC_)                                                                        \
    do {                                                                                                \
        a.tiles_x = ic_cdiv(W, TC_); a.tiles_y = ic_cdiv(H, TR_);                                       \
        if (g_pipe < 0 ? (PT_ >= 2) : g_pipe)                                                           \
            hipLaunchKernelGGL((conv3x3_c128_pipe_kernel<PT_, TR_, TC_>), dim3(a.tiles_x * a.tiles_y * N), \
                               dim3(256), g_lds_pad, (hipStream_t)stream, a);                           \
        else                                                                                            \
            hipLaunchKernelGGL((conv3x3_c128_kernel<PT_, TR_, TC_>), dim3(a.tiles_x * a.tiles_y * N),   \
                               dim3(256), g_lds_pad, (hipStream_t)stream, a);                           \
    } while (0)

extern "C" int ic_conv3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale,
                                          const float* shift, const float* res1, const float* res2, float* y,
                                          int N, int H, int W, int relu, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y);
    IC_CHECK_ARG(N > 0 && H > 0 && W > 0);
    if ((long long)C128 * H * W >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    C3Args a{};
    a.x = x; a.wp = w_packed; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.relu = relu; a.dbg = g_dbg; a.abl = g_abl;
    switch (pick_variant(N, H, W, flags)) {
        case 0: C3_LAUNCH(4, 8, 16); break;
        case 1: C3_LAUNCH(4, 4, 32); break;
        case 2: C3_LAUNCH(3, 8, 12); break;
        case 3: C3_LAUNCH(3, 6, 16); break;
        case 4: C3_LAUNCH(3, 3, 32); break;
        case 5: C3_LAUNCH(2, 4, 16); break;
        case 6: C3_LAUNCH(2, 2, 32); break;
        case 7: C3_LAUNCH(2, 8, 8); break;
        case 8: C3_LAUNCH(1, 2, 16); break;
        default: C3_LAUNCH(1, 4, 8); break;
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}
