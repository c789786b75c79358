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
//      (double buffered, register-staged so the global loads of chunk c+1 fly under the MFMAs of
//      chunk c); all 9 taps and all 4 waves re-read it from LDS with conflict-free ds_read_b32.
// Work-group = 256 threads = 4 waves; wave w owns output channels [32w, 32w+32) for all TRxTC pixels
// of the tile (PT = TR*TC/32 accumulator tiles of 32x32 = 16*PT accumulator registers).
// K order per output: chunk (8 ci) -> tap (ky,kx) -> k-step (2 ci) ; fixed, independent of the tile
// position, so results do not depend on the launch geometry.
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
};

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

template <int PT, int TR, int TC>
__global__ __launch_bounds__(256) void conv3x3_c128_kernel(const C3Args a) {
    static_assert(TR * TC == 32 * PT, "tile must be PT MFMA pixel tiles");
    constexpr int S = TC + 2;               // LDS row stride (floats)
    constexpr int CS = (TR + 2) * S;        // LDS channel stride
    constexpr int CHUNK = KC * CS;          // floats per staged chunk
    constexpr int NST = (CHUNK + 255) / 256;
    __shared__ float lds[2][NST * 256];    // padded so that staging writes need no bounds branch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; const int n = b / a.tiles_y;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * C128 * HW;

    // ---- staging plan: element e = tid + 256 i of the chunk [ci][row][col] ----
    int goff[NST];
    unsigned inb = 0;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / CS, rem = e - ci * CS;
        const int r = rem / S, c = rem - r * S;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = (e < CHUNK) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[i] = ok ? ci * HW + gy * a.W + gx : 0;
        inb |= (ok ? 1u : 0u) << i;
    }

    // ---- B-operand (pixel) read offsets per accumulator tile ----
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

    // ---- prologue: chunk 0 ----
    f32x4 wcur[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wcur[t] = wp[t * 256];
    {
        float st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) { const float v = xin[goff[i]]; st[i] = ((inb >> i) & 1) ? v : 0.f; }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            lds[0][tid + 256 * i] = st[i];
        }
    }
    __syncthreads();

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
            for (int i = 0; i < NST; ++i) {
                lds[buf ^ 1][tid + 256 * i] = st[i];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) wcur[t] = wnext[t];
        }
        __syncthreads();
    }

    // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel), j = lane&31 (pixel) ----
    const size_t obase = (size_t)n * C128 * HW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        const int oy = y0 + q / TC, ox = x0 + q % TC;
        if (oy < a.H && ox < a.W) {
            const size_t pix = obase + (size_t)oy * a.W + ox;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = fmaf(acc[p][r], a.scale[co], a.shift[co]);
                if (a.relu) v = fmaxf(v, 0.f);
                const size_t o = pix + (size_t)co * HW;
                if (a.res1) v += a.res1[o];
                if (a.res2) v += a.res2[o];
                a.y[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant of the same computation (identical K order -> identical results).
//  * the B operands of tap t+1 are fetched from LDS while the MFMAs of tap t issue (register double
//    buffer; one ds_read slotted behind every MFMA by sched_group_barrier) -> every LDS read has 4*PT
//    MFMA issue slots (>= 256 cycles) to land, so a single wave per SIMD keeps the matrix pipe fed;
//  * the LDS ring is 3 deep and the hand-over barrier sits in the MIDDLE of a chunk: chunk c+1 is
//    written (and the barrier passed) by tap 4 of chunk c, so the last tap of chunk c can already
//    prefetch tap 0 of chunk c+1 and the MFMA stream never drains at a chunk boundary.
//    (WAR: buffer (c+1)%3 was last read in chunk c-2; every wave has since passed chunk c-1's barrier.)
template <int PT, int TR, int TC>
__global__ __launch_bounds__(256) void conv3x3_c128_pipe_kernel(const C3Args a) {
    static_assert(TR * TC == 32 * PT, "tile must be PT MFMA pixel tiles");
    constexpr int S = TC + 2;
    constexpr int CS = (TR + 2) * S;
    constexpr int CHUNK = KC * CS;
    constexpr int NST = (CHUNK + 255) / 256;
    __shared__ float lds[3][NST * 256];    // padded so that staging writes need no bounds branch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y; const int n = b / a.tiles_y;
    const int x0 = tx * TC, y0 = ty * TR;
    const int HW = a.H * a.W;
    const float* __restrict__ xin = a.x + (size_t)n * C128 * HW;

    int goff[NST];
    unsigned inb = 0;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = tid + 256 * i;
        const int ci = e / CS, rem = e - ci * CS;
        const int r = rem / S, c = rem - r * S;
        const int gy = y0 + r - 1, gx = x0 + c - 1;
        const bool ok = (e < CHUNK) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[i] = ok ? ci * HW + gy * a.W + gx : 0;
        inb |= (ok ? 1u : 0u) << i;
    }
    const int j = lane & 31, kh = lane >> 5;
    int boff[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        boff[p] = kh * CS + (q / TC) * S + (q % TC);
    }
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.wp) + wave * 64 + lane;

    f32x16 acc[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x4 wcur[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wcur[t] = wp[t * 256];
    {
        float st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) { const float v = xin[goff[i]]; st[i] = ((inb >> i) & 1) ? v : 0.f; }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            lds[0][tid + 256 * i] = st[i];
        }
    }
    __syncthreads();

    float bq[2][4][PT];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int p = 0; p < PT; ++p) bq[0][ks][p] = lds[0][boff[p] + 2 * ks * CS];

    int ring = 0;                                   // c % 3
    for (int c2 = 0; c2 < NCHUNK; c2 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {               // two chunks per trip: register parity stays static
            const int c = c2 + h;
            const bool more = c + 1 < NCHUNK;
            const int ringn = ring == 2 ? 0 : ring + 1;
            const float* __restrict__ L = lds[ring];
            float* __restrict__ Ln = lds[ringn];
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
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int cur = (h * 9 + t) & 1;
                if (t == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int i = 0; i < NST; ++i) {
                            Ln[tid + 256 * i] = st[i];
                        }
                    }
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
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
                    const float av = wcur[t][ks];
#pragma unroll
                    for (int p = 0; p < PT; ++p)
                        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[cur][ks][p], acc[p], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4 * PT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
            }
            if (more) {
#pragma unroll
                for (int t = 0; t < 9; ++t) wcur[t] = wnext[t];
            }
            ring = ringn;
        }
    }

    const size_t obase = (size_t)n * C128 * HW;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int q = 32 * p + j;
        const int oy = y0 + q / TC, ox = x0 + q % TC;
        if (oy < a.H && ox < a.W) {
            const size_t pix = obase + (size_t)oy * a.W + ox;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh;
                float v = fmaf(acc[p][r], a.scale[co], a.shift[co]);
                if (a.relu) v = fmaxf(v, 0.f);
                const size_t o = pix + (size_t)co * HW;
                if (a.res1) v += a.res1[o];
                if (a.res2) v += a.res2[o];
                a.y[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct C3Variant { int PT, TR, TC; };
static const C3Variant kVariants[] = {
    {4, 8, 16}, {4, 4, 32}, {3, 8, 12}, {3, 6, 16}, {3, 3, 32}, {2, 4, 16}, {2, 2, 32}, {2, 8, 8}, {1, 4, 8}, {1, 2, 16},
};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
static int g_variant_override = -1;
static int g_lds_pad = 0;      // extra dynamic LDS per work-group (tuning: limits work-groups per CU)
static int g_pipe = 0;         // 0: compiler-scheduled inner loop, 1: explicit register double buffer

extern "C" int ic_conv3x3_c128_set_variant(int v) {
    int prev = g_variant_override;
    g_variant_override = (v >= 0 && v < kNumVariants) ? v : -1;
    return prev;
}

extern "C" int ic_conv3x3_c128_set_tuning(int key, int value) {
    int prev = -1;
    switch (key) {
        case 0: return ic_conv3x3_c128_set_variant(value);
        case 1: prev = g_lds_pad; g_lds_pad = value < 0 ? 0 : value; break;
        case 2: prev = g_pipe; g_pipe = value ? 1 : 0; break;
        default: break;
    }
    return prev;
}

static int pick_variant(int N, int H, int W) {
    if (g_variant_override >= 0) return g_variant_override;
    int best = 0; double bestc = 1e30;
    for (int v = 0; v < kNumVariants; ++v) {
        const C3Variant& k = kVariants[v];
        const long nwg = (long)N * ic_cdiv(H, k.TR) * ic_cdiv(W, k.TC);
        // one work-group per CU per round; fixed per-group cost (prologue, epilogue) ~ 0.3 tile units
        const double cost = (double)((nwg + 255) / 256) * (k.PT + 0.3);
        if (cost < bestc - 1e-9) { bestc = cost; best = v; }
    }
    return best;
}

extern "C" size_t ic_conv3x3_c128_packed_floats(void) { return PACKED_FLOATS; }

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
        if (g_pipe)                                                                                     \
            hipLaunchKernelGGL((conv3x3_c128_pipe_kernel<PT_, TR_, TC_>), dim3(a.tiles_x * a.tiles_y * N), \
                               dim3(256), g_lds_pad, (hipStream_t)stream, a);                           \
        else                                                                                            \
            hipLaunchKernelGGL((conv3x3_c128_kernel<PT_, TR_, TC_>), dim3(a.tiles_x * a.tiles_y * N), \
                               dim3(256), g_lds_pad, (hipStream_t)stream, a);                           \
    } while (0)

extern "C" int ic_conv3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale,
                                          const float* shift, const float* res1, const float* res2, float* y,
                                          int N, int H, int W, int relu, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y);
    IC_CHECK_ARG(N > 0 && H > 0 && W > 0);
    if ((long long)C128 * H * W >= (1ll << 31)) return IC_ERR_UNSUPPORTED;
    C3Args a{};
    a.x = x; a.wp = w_packed; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    switch (pick_variant(N, H, W)) {
        case 0: C3_LAUNCH(4, 8, 16); break;
        case 1: C3_LAUNCH(4, 4, 32); break;
        case 2: C3_LAUNCH(3, 8, 12); break;
        case 3: C3_LAUNCH(3, 6, 16); break;
        case 4: C3_LAUNCH(3, 3, 32); break;
        case 5: C3_LAUNCH(2, 4, 16); break;
        case 6: C3_LAUNCH(2, 2, 32); break;
        case 7: C3_LAUNCH(2, 8, 8); break;
        case 8: C3_LAUNCH(1, 4, 8); break;
        default: C3_LAUNCH(1, 2, 16); break;
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}
