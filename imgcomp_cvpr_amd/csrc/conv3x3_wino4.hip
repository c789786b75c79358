// 3x3, stride 1, 128 -> 128 channel convolution of the residual stacks (autoencoder.py:224-234, :252-262, :274-287) in Winograd
// F(4x4, 3x3) form on the fp32 matrix cores: 36 multiplies per 16 outputs -- 36 / 144 of the direct form's, 0.5625 of F(2x2,3x3)'s.
//
//   Y = At [ (G g Gt) (.) (Bt d B) ] A        g: 3x3 filter, d: 6x6 input patch, Y: 4x4 outputs ("tile"); points 0, +-1, +-2, inf
//
// Numerics first (tools/wino_f4_numerics.py, round 4): the whole encoder / decoder evaluated in float32 with this form has the SAME
// error against the float64 oracle as with the direct form or F(2x2) -- z 1.1-1.6e-5, x_out 3e-7 of the tensor scale, no symbol
// flips: the network's error is set by its other parts, the larger transform constants do not show.  (Round 2 had rejected the
// form by scaling the device's whole-network error with a per-layer factor; the per-layer factor does not carry through
// BatchNorm + skips.)  One layer alone: 4-8e-6 of the tensor scale against 3-5e-7 for F(2x2) (tests/test_gpu_ops.py, bound 2e-5).
//
// Layout on a gfx950 wave, v_mfma_f32_16x16x4_f32 (D[16 x 16] += A[16 x 4] B[4 x 16]):
//   * M axis = 16 output channels, N axis = 16 tiles (a SEGMENT: 16 horizontally adjacent tiles = 4 x 64 output pixels),
//     K = 4 input channels.  Lane l = (tile n = l & 15, k = l >> 4): ONE 6x6 patch transform per lane yields its B operands of
//     all 36 positions of a k-step; a wave owns 16 channels x 16 tiles x 36 positions = 36 accumulators of 4 registers = 144
//     registers (a 32 x 32 tile would need 576), so TWO work-groups fit a CU and each SIMD always has a second wave to issue from.
//   * the output transform At M A is lane-local (same lane, same register index across the 36 accumulators): 4 channels x 4 x 4
//     pixels per lane, stored as 16-byte runs (a tile row is 4 neighbouring pixels; 16 lanes = 256 contiguous bytes).
//   * work-group = 4 waves = 4 channel tiles (one HALF of the output channels) of one segment.  Wave w loads and transforms the
//     input of k-steps 4 j + w only; the B operands reach the other waves through a two-half LDS ring (72 KB), one barrier per
//     4 k-steps.  Patch columns: every lane loads its own aligned 4 pixels per patch row, the two outer columns are the neighbour
//     lanes' values (DPP row shifts; the two ends of the 16-lane tile row load one extra dword).
//   * A operands (transformed filters, 36 x 128 x 128 floats = 2.36 MB per layer) are pre-packed in fragment order: 9 16-byte
//     loads per k-step and lane, each 1 KB contiguous across the wave, L2 resident, streamed through a register ring.
//   * zero padding is done by the memory system (raw buffer loads, out-of-range lane offsets), as in the F(2x2) kernels.
// Shapes: W % 4 == 0 (aligned 16-byte rows); anything else keeps the F(2x2) forms (ic_conv3x3_c128_auto_f32).
//
// Where the time goes (in-kernel shader-clock stamps, a build with -DW4_STAMPS, tools/w4prof.py; batch of 8 Kodak-sized maps, two
// work-groups per CU, ~1.9 GHz under this load): per wave prologue 9.5 k clocks, loop 89 k, epilogue 8.5 k, against 36.9 k clocks of
// MFMA issue for its 1152 instructions -- two waves share a SIMD's matrix pipe, so 73.7 k of every 107 k (0.69).  Alone on its
// SIMD a wave's loop takes 56 k.  How it got there (round 4, same launch): first correct version 201 us; filter ring 4 -> 6 quads,
// patch requested ahead of its transform 189 us; the transform of a turn cut into 26 pieces behind MFMAs (loop alone 75 k -> 57 k
// clocks; no change at full load, where the other wave fills the pipe anyway); epilogue operands two channels ahead and in front
// of the stores (epilogue 20 k -> 8.5 k clocks) 175 us.  F(2x2) plan on the same input: 242 us.
//
// INLINE-ASM MFMAs AND THE COMPILER (round 5; experiment log: profiles/r05_w4_rootcause.md).  The MFMAs below are asm statements so that
// the accumulator stays tied to the destination.  hipcc orders an asm statement by its operands but does not know it is an MFMA, so it
// pads none of the matrix-pipe hazards the hardware leaves to software:
//   (A) a non-MFMA instruction touching an MFMA's destination needs >= 11 wait states behind it (8-pass XDL; hipcc: s_nop 9 + 1),
//   (B) a VALU write of a register an MFMA reads needs 2 wait states before it (hipcc: s_nop 1).
// The code written here keeps both by construction: A / B operands come from LDS and buffer loads (waited for, no wait states
// needed), accumulators are touched only behind the `s_nop 15 x 2` pad after the loop.  What it cannot control is the REGISTER
// ALLOCATOR: under more pressure -- round 4: the SLP vectoriser's 64-bit temporaries in the input transform -- it spills and copies
// the four VGPR-resident accumulators (scratch_store_dwordx4 / v_mov_b64) directly behind and in front of the MFMAs that own
// them.  That was round 4's "packed fp32" failure: wrong values in the columns 12..15 of 16-lane rows (the part of an MFMA result
// that lands last), a different place at every launch, only with a second wave on the SIMD competing for the matrix pipe.  Proof:
// s_nops for (A) and (B) patched into the failing build's assembly make it bit-exact (0 wrong launches of 150 against 150 of 150;
// (A) alone leaves 3e-4 of the errors, (B) alone all of them; packed-op, DPP, ds_write and SrcA/B write-after-read spacing change
// nothing).  The packed instructions themselves are innocent: today's source built WITH the vectoriser has no such spill and no
// failure.  Guard: csrc/isa_audit.py checks the assembly of this file for (A), (B) at every build (csrc/Makefile) and fails it on
// a finding -- whatever flags or compiler made it; -fno-slp-vectorize stays as a speed choice (packed fp32 next to MFMAs costs
// issue slots and registers).  Run-time half: tests/test_gpu_ops.py::test_conv3x3_c128_winograd_f4_full_load_is_deterministic (every
// instantiation, two work-groups per CU) and tests/test_gpu_bench.py::test_in_flight_schedule_is_bit_identical_to_serial.
#include "wino_common.h"
#include "internal.h"

#define W4_QUADS 9                                  // 36 positions in quads of 4
#define W4_PACKED_FLOATS (36 * WN_C * WN_C)
#define W4_ACC_A 32                                 // accumulators (of 36) kept in AGPRs
// tuning knobs (defaults = the measured best, round 4: 4 waves, round 5: rings 9 / 3, turn at quad 6, transform 10 quads after its request)
#ifndef W4_RA
#define W4_RA 9                                     // filter-fragment ring, in quads (36 % W4_RA == 0); round 5: 6 -> 9 (below)
#endif
#ifndef W4_RB
#define W4_RB 3                                     // B-operand ring, in quads (36 % W4_RB == 0)
#endif
#ifndef W4_TURN
#define W4_TURN 6                                   // quad of an iteration at which a wave requests its k-step of the next one
#endif
#ifndef W4_GAP
#define W4_GAP 10                                   // quads between that request and the transform (round 5: 5 -> 10: a lone wave has nobody to cover
                                                    //    the patch's round trip; one image at a time 171.1 -> 175.6 Mpix/s, four in flight unchanged; 8, 14, 16, 22 the same)
#endif
#ifndef W4_PRE_N
#define W4_PRE_N 2                                  // channels whose residual 1 is requested that early (the other of the first two: at the epilogue's start)
#endif
#ifndef W4_PRE
#define W4_PRE 31                                   // quad of the LAST iteration at which the epilogue's first operands are requested (-1: in the epilogue)
#endif
#ifndef W4_WT_MAX
#define W4_WT_MAX 512                                // launches of at most this many work-groups (a single round) store write-through
#endif
#ifndef W4_ABL
#define W4_ABL 0                                    // timing-only ablations (wrong values; never in the shipped library): see the loop
#endif
#ifndef W4_SLICE1
#define W4_SLICE1 1                                 // 1: both parts of a transform slice behind ONE MFMA of the quad: 13 gaps of ~12 instructions per turn instead of
                                                    //    26 of ~6 (round 5: one image at a time 171.2 against 169.3 Mpix/s, in flight equal)
#endif
#ifndef W4_SOFT
#define W4_SOFT 0                                   // 1: two barriers per iteration between MFMA quads (B1 / B2 below); 0: one at its end (measured equal: below)
#endif
#ifndef W4_WG8_FLAGS
#define W4_WG8_FLAGS 1                              // 8-wave form: 1 = the ring's two hazards are kept by LDS sequence counters (no barrier inside the loop), 0 = one 512-thread barrier per iteration
#endif
#define W4_WG8_SPINS (1 << 16)                      // bound of a counter poll (~100 clocks each): a miscount gives wrong values, never a hung GPU
#define W4_B1 29
#define W4_B2 8
#ifndef W4_SPREAD
#define W4_SPREAD 1                                 // 1: the transform of a turn is cut into 26 pieces of 6 vector instructions, two per
#endif                                              //    quad, each behind an MFMA (whose 8 passes hide them); 0: one block of ~200

// ---- filter transform + packing: U = G g Gt in float64, rounded once -------------------------------------------------------
// packed float index: (((cot * KS + ks) * 9 + p / 4) * 64 + lane) * 4 + p % 4, cot = co / 16, ks = ci / 4 (KS = Cin / 4 k-steps),
// lane = (ci & 3) * 16 + (co & 15), p = 6 xi + nu.  What the 3x3 filter g[a][b][ci][co] is (mode):
//   0  the layer's own [3][3][128][128] array
//   1  its adjoint (data gradient): g[a][b][in = co][out = ci] = w[2 - a][2 - b][ci][co]
//   2  a 5x5 / stride-2 SAME convolution [5][5][64][128] (h2, autoencoder.py:223) as ONE 3x3 convolution over the four phases of its
//      input stacked as 256 channels (ci = (2 py + px) * 64 + c):  Y[i] = sum_a W[a] X[2 i + a - 1], so the odd rows X[2 m + 1] meet
//      the taps (W[0], W[2], W[4]) at m = i - 1, i, i + 1 and the even rows X[2 m] the taps (0, W[1], W[3])
//   3  the 5x5 / stride-2 transposed convolution [5][5][64 out][128 in] (h12, autoencoder.py:264) as ONE 3x3 convolution to four
//      phase planes per output channel (co = 4 c + 2 py + px): output rows 2 m + 1 take (W[4], W[2], W[0]) at input rows
//      m - 1, m, m + 1, output rows 2 m take (W[3], W[1], 0)
// (tools/phase_conv_check.py checks both identities against the oracle's convolutions in float64.)
__device__ __forceinline__ int w4_tap5(int mode, int phase, int a) {          // 5-tap index of 3-tap position a for a phase, -1: none
    if (mode == 2) return phase ? 2 * a : (a == 0 ? -1 : 2 * a - 1);
    return phase ? 4 - 2 * a : (a == 2 ? -1 : 3 - 2 * a);
}
// (blockIdx.y = layer of a batch: w_tab, when given, holds every layer's filter pointer and the fragments follow each other in out)
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w_tf, const float* const* __restrict__ w_tab, float* __restrict__ out,
                                                        int mode, int CI, int CO) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CI * CO) return;
    if (w_tab) { w_tf = w_tab[blockIdx.y]; out += (size_t)blockIdx.y * 36 * CI * CO; }
    const int cin = idx / CO, cout = idx % CO;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (mode == 0) g[a][b] = (double)w_tf[((a * 3 + b) * WN_C + cin) * WN_C + cout];
            else if (mode == 1) g[a][b] = (double)w_tf[(((2 - a) * 3 + (2 - b)) * WN_C + cout) * WN_C + cin];
            else if (mode == 2) {
                const int ph = cin >> 6, c = cin & 63, ky = w4_tap5(2, ph >> 1, a), kx = w4_tap5(2, ph & 1, b);
                g[a][b] = (ky < 0 || kx < 0) ? 0.0 : (double)w_tf[((ky * 5 + kx) * 64 + c) * 128 + cout];
            } else {
                const int ph = cout & 3, c = cout >> 2, ky = w4_tap5(3, ph >> 1, a), kx = w4_tap5(3, ph & 1, b);
                g[a][b] = (ky < 0 || kx < 0) ? 0.0 : (double)w_tf[((ky * 5 + kx) * 64 + c) * 128 + cin];
            }
        }
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    double t[6][3];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[x][b] = G[x][0] * g[0][b] + G[x][1] * g[1][b] + G[x][2] * g[2][b];
    float* o = out + ((size_t)((cout >> 4) * (CI >> 2) + (cin >> 2)) * W4_QUADS * 64 + ((cin & 3) * 16 + (cout & 15))) * 4;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int v = 0; v < 6; ++v) {
            const int p = 6 * x + v;
            o[(p >> 2) * 256 + (p & 3)] = (float)(t[x][0] * G[v][0] + t[x][1] * G[v][1] + t[x][2] * G[v][2]);
        }
}

extern "C" size_t ic_wino4_3x3_c128_packed_floats(void) { return W4_PACKED_FLOATS; }

extern "C" int ic_pack_wino4_3x3_c128_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(WN_C * WN_C / 256), dim3(256), 0, (hipStream_t)stream, w_tf, nullptr, w_packed, backward ? 1 : 0, WN_C, WN_C);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// every 3x3 filter of a network in one launch (training re-packs them at every step): w_tf_table_dev[l] -> w_packed + l * packed_floats
extern "C" int ic_pack_wino4_3x3_c128_batch_f32(const float* const* w_tf_table_dev, float* w_packed, int layers, int backward, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf_table_dev && w_packed && layers > 0);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(WN_C * WN_C / 256, layers), dim3(256), 0, (hipStream_t)stream, nullptr, w_tf_table_dev, w_packed,
                       backward ? 1 : 0, WN_C, WN_C);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// h2 / h12 in phase form: 36 x 256 x 128 fragments either way
extern "C" size_t ic_wino4_conv5s2_packed_floats(void) { return (size_t)36 * 256 * 128; }

extern "C" int ic_pack_wino4_conv5s2_f32(const float* w_tf, float* w_packed, int transposed, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3(256 * 128 / 256), dim3(256), 0, (hipStream_t)stream, w_tf, nullptr, w_packed, transposed ? 3 : 2,
                       transposed ? 128 : 256, transposed ? 256 : 128);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// Bt x of 6 values (points 0, +-1, +-2, inf):
//   t0 = 4 d0 - 5 d2 + d4      t1 = -4 d1 - 4 d2 + d3 + d4      t2 = 4 d1 - 4 d2 - d3 + d4
//   t3 = -2 d1 - d2 + 2 d3 + d4      t4 = 2 d1 - d2 - 2 d3 + d4      t5 = 4 d1 - 5 d3 + d5
// (inputs by value: the outputs may be the same variables)
__device__ __forceinline__ void w4_bt(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                      float& t0, float& t1, float& t2, float& t3, float& t4, float& t5) {
    const float p = fmaf(-4.f, d2, d4), q = fmaf(-4.f, d1, d3);       // d4 - 4 d2, d3 - 4 d1
    const float r = d4 - d2, s = 2.f * (d3 - d1);
    const float a0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    const float a5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
    t0 = a0;
    t1 = p + q;
    t2 = p - q;
    t3 = r + s;
    t4 = r - s;
    t5 = a5;
}
// At x of 6 values: y0 = m0 + m1 + m2 + m3 + m4, y1 = (m1 - m2) + 2 (m3 - m4), y2 = (m1 + m2) + 4 (m3 + m4), y3 = (m1 - m2) + 8 (m3 - m4) + m5
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5,
                                      float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = (m0 + s1) + s2;
    y1 = fmaf(2.f, d2, d1);
    y2 = fmaf(4.f, s2, s1);
    y3 = fmaf(8.f, d2, d1) + m5;
}

// (Variants built, measured and removed in rounds 4-5 -- a one-wave-per-SIMD build with 18 / 6 rings, four producer waves per work-group,
// an 8-wave K-split work-group, round 4's barrier-synchronised 8-wave work-group: docs/history/DESIGN_section3_rounds_1_5.md, appendix.
// Round 6's 8-wave form with LDS counters is the WG8 template parameter below; profiles/r06_w4_wg8.md has its measurements.)
// A work-group (4 waves) is one HALF of the output channels of a segment, two work-groups per CU.
// Template: WT write-through stores (single-round launches); RES how many residual inputs the epilogue serves (0, 1, 2); CIN / COUT
// channels (128 / 128: the residual layers; 256 / 128: h2 over its input's phases; 128 / 256: h12 to its output's phases); SHUF:
// the four "channels" of a lane are the four phases of ONE output channel and are stored interleaved into the 2 H x 2 W map.
// SEG2: the 16 tiles of a segment are 2 rows x 8 columns (lanes 0..7 / 8..15 of a 16-lane row) instead of 1 x 16 -- maps narrower than
// 16 tiles (training crops: 32 x 32 maps = 8 x 8 tiles) then fill their segments.  Row validity and row offsets become per-lane (two
// values per wave: scalar conditions combined with the lane's half), and the lanes 7 / 8 in the middle of a DPP row take the column
// outside from their own edge load like the lanes 0 / 15 at its ends.
// WG8 (round 6): ONE work-group of 8 waves covers all 128 output channels of a segment (COUT / 128 work-groups per segment), one per CU.
// Waves w and w + 4 share a SIMD.  The input transform is made ONCE per segment: the waves 0..3 (group 0) make the k-steps of the even
// iterations, the waves 4..7 (group 1) those of the odd ones -- every SIMD has one transforming and one MFMA-only wave in every iteration.
// Same ring, same barrier (now 512 threads), same values in the same order: bit-identical to the 4-wave form.
// STATS (round 6, the training step's forward convolutions): the RAW convolution is stored (no BatchNorm fold, activation or residual)
// and the epilogue also leaves, per output channel and segment, the sum and the sum of squares of the values it stores -- the batch
// statistics of training-mode BatchNorm (autoencoder.py:106-125) without a pass over the tensor: ic_bn_train_forward_cstats_f32 folds them.
template <bool WT, int RES, int CIN, int COUT, bool SHUF, bool SEG2, bool WG8, bool STATS = false>
__global__ __launch_bounds__(WG8 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wino4_3x3_kernel(const WnArgs a) {
    constexpr int WAVES = WG8 ? 8 : 4;
    constexpr int KS = CIN / 4, IT = KS / 4, PARTS = COUT / (16 * WAVES);   // k-steps, iterations of 4 k-steps, work-groups per segment
    __shared__ f32x4 ring[2 * 4 * W4_QUADS * 64];                 // [half][k-step of the iteration][position quad][lane]: 72 KB
    __shared__ float pf_sink[64];                                 // where the prefetch below lands (never read).  It sits BEHIND the 72 KB ring: the LDS-DMA base in M0
                                                                  // is above 64 KB, which gfx950 honours (a part that kept 16 bits would sink into ring half 0);
                                                                  // tests/test_gpu_network.py::test_next_layer_filter_prefetch_changes_no_bit compares prefetch on / off bit for bit
    __shared__ unsigned w8_flag[4];                               // WG8: [h] = producer waves that completed ring half h, [2 + h] = waves that finished reading it (running totals)
#ifdef W4_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
#ifdef W4_LAUNCH_STAMPS
    const unsigned long long ls_c0 = __builtin_amdgcn_s_memtime(), ls_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, n16 = lane & 15, kq = lane >> 4;
    const int b = a.xcd_runs ? ic_xcd_run(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int part = b % PARTS;
    const int seg = b / PARTS + a.g0;
    const int sx = seg % a.gcols, t_ = seg / a.gcols;
    const int ty = t_ % a.grows, n = t_ / a.grows;
    const int cot = part * WAVES + wave;                          // 16-channel tile of this wave
    const int pw = wave & 3;                                      // k-step of an iteration this wave produces
    const int grp = WG8 ? wave >> 2 : 0;                          // WG8: this wave transforms the iterations of this parity (scalar)
    const int hrow = SEG2 ? (n16 >> 3) : 0;                       // SEG2: which of the segment's two tile rows this lane works on
    const int tx = SEG2 ? 8 * sx + (n16 & 7) : 16 * sx + n16;
    const int H = a.H, W = a.W, HW = H * W;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)n * CIN * HW), 0, CIN * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 36 * CIN * COUT * 4, 0x00020000);
    // patch rows 4 ty - 1 .. 4 ty + 4: own aligned 4 pixels (columns 4 tx .. 4 tx + 3); the end lanes of a tile row also fetch the
    // column outside (first lane: 4 tx - 1, last lane: 4 tx + 4), every other lane gets an out-of-range offset there.
    // Row validity is wave-uniform (SEG2: one value per half), column validity per lane: ONE lane offset for the patch's first row
    // (out of range when the tile lies beyond the map) and one for the end column; a row outside the image swaps in the
    // out-of-range offset by a scalar condition.
    const bool col_ok = 4 * tx < W;
    const bool first_lane = SEG2 ? (n16 & 7) == 0 : n16 == 0, last_lane = SEG2 ? (n16 & 7) == 7 : n16 == 15;
    const bool mid_l = SEG2 && n16 == 8, mid_r = SEG2 && n16 == 7;       // lanes whose DPP neighbour belongs to the other tile row
    const int ecol = first_lane ? 4 * tx - 1 : (last_lane ? 4 * tx + 4 : -1);
    const bool e_ok = col_ok && ecol >= 0 && ecol < W;
    const int tyr = SEG2 ? 2 * ty : ty;                           // (first) tile row of the segment: scalar
    const int r_first = 4 * tyr - 1;
    // (offsets are formed in int: r_first may be -1; the rows actually used are >= 0)
    const int obase = (kq * HW + (r_first + 4 * hrow) * W + 4 * tx) * 4;
    const int ebase = (kq * HW + (r_first + 4 * hrow) * W + ecol) * 4;
    const unsigned fo = (unsigned)lane * 16u;

    // Accumulators are tied to the MFMA's destination by inline asm: the builtin lets the allocator put the result into ANOTHER
    // tuple than the addend, which doubles the accumulator footprint for the duration and pushes the rings into scratch.
    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (at two waves per SIMD the compiler splits the 256 registers of a wave 128 : 128 between the two files: 32 of the 36
    // accumulators sit in AGPRs, the last four in VGPRs -- the MFMA takes its addend from either file)
#pragma unroll
    for (int p = 0; p < 36; ++p) {
        if (p < W4_ACC_A) asm volatile("" : "+a"(acc[p]));
        else asm volatile("" : "+v"(acc[p]));
    }
    __builtin_amdgcn_sched_barrier(0);

    // B-operand reads of ring half 1 (byte offsets 0x9000 .. 0x12000): a ds_read's immediate offset has 16 bits, so the compiler keeps
    // one address register per 1 KB quad beyond 0x10000 -- eight of them through the whole loop (the 4-wave form has the room, the
    // 8-wave form spilled five).  WG8: ONE opaque base register at the start of half 1 instead.
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    unsigned rb1 = (unsigned)(size_t)&ring[4 * W4_QUADS * 64 + lane];
    if (WG8) asm volatile("" : "+v"(rb1));
    auto ring_rd = [&](int half, int idx) __attribute__((always_inline)) -> f32x4 {      // ring[(half * 4 * W4_QUADS + idx) * 64 + lane]
        if (WG8 && half) return *(reinterpret_cast<lds_f32x4*>((size_t)rb1) + idx * 64);
        return ring[(half * 4 * W4_QUADS + idx) * 64 + lane];
    };

    f32x4 pr[6];            // own 4 pixels of the 6 patch rows
    float pe[6];            // the column outside (lanes 0 / 15 of a tile row)
    auto load_patch = [&](int ks) __attribute__((always_inline)) {
        const int so = ks * 4 * HW * 4;                           // scalar: channels 4 ks ..
        // the 12 lane offsets are re-derived from two registers at every call: hoisted out of the loop (which the compiler does
        // on its own) they would sit in 12 registers next to 144 accumulators and spill
        int ob = obase, eb = ebase;
        asm volatile("" : "+v"(ob), "+v"(eb));
        // (Round 5: a second path for segments whose six patch rows all lie inside the image -- row steps in the SCALAR offset, the two
        // lane offsets carrying only the column's validity, no vector instruction per load instead of an add and a select -- was
        // slower: one image at a time 163.2 against 171.0 Mpix/s, 8 Kodak maps 183.3 against 179.7 us.  Removed.)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const bool row_lo = r_first + i >= 0 && r_first + i < H;          // scalar
            const bool row_hi = r_first + 4 + i < H;                          // scalar: the same patch row of the segment's second tile row
            const bool row_ok = SEG2 ? (hrow ? row_hi : row_lo) : row_lo;
            const unsigned o1 = (row_ok && col_ok) ? (unsigned)(ob + i * W * 4) : WN_OOB;
            const unsigned o2 = (row_ok && e_ok) ? (unsigned)(eb + i * W * 4) : WN_OOB;
            pr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, o1, so, 0));
            pe[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o2, so, 0));
        }
    };
    // Bt d B of this lane's patch -> ring[half][st = wave][quad][lane].  In place on 36 registers: columns first (each 6 -> 6),
    // then rows, a pair of rows = three position quads written as soon as it is complete; the phases are fenced so that the
    // scheduler does not interleave them (36 accumulators + rings leave ~50 registers for all of this)
    auto transform_put = [&](int half) __attribute__((always_inline)) {
        float u[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            u[i][0] = dpp_from_left(pe[i], pr[i][3]);            // column 4 tx - 1 = the left neighbour's last pixel
            u[i][5] = dpp_from_right(pe[i], pr[i][0]);           // column 4 tx + 4 = the right neighbour's first pixel
            if (SEG2) { u[i][0] = mid_l ? pe[i] : u[i][0]; u[i][5] = mid_r ? pe[i] : u[i][5]; }     // the row's middle is an end too
            u[i][1] = pr[i][0]; u[i][2] = pr[i][1]; u[i][3] = pr[i][2]; u[i][4] = pr[i][3];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            w4_bt(u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j], u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4* dst = &ring[((half * 4 + pw) * W4_QUADS) * 64 + lane];
#pragma unroll
        for (int x = 0; x < 6; x += 2) {
            float v0[6], v1[6];
            w4_bt(u[x][0], u[x][1], u[x][2], u[x][3], u[x][4], u[x][5], v0[0], v0[1], v0[2], v0[3], v0[4], v0[5]);
            w4_bt(u[x + 1][0], u[x + 1][1], u[x + 1][2], u[x + 1][3], u[x + 1][4], u[x + 1][5], v1[0], v1[1], v1[2], v1[3], v1[4], v1[5]);
            const int q0 = 3 * (x / 2);                           // positions 6 x .. 6 x + 11 = quads q0 .. q0 + 2
            dst[(q0 + 0) * 64] = f32x4{v0[0], v0[1], v0[2], v0[3]};
            dst[(q0 + 1) * 64] = f32x4{v0[4], v0[5], v1[0], v1[1]};
            dst[(q0 + 2) * 64] = f32x4{v1[2], v1[3], v1[4], v1[5]};
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // The same transform cut into pieces for the loop (W4_SPREAD): slice k = 0 .. 12 of a turn, two parts each, a part = 6 vector
    // instructions placed directly behind an MFMA -- the matrix pipe works 8 passes (32 clocks) on it, the vector unit is free for 7
    // instructions of the same wave meanwhile.  As one block the ~200 instructions of a turn stop the wave's MFMA issue for ~2 k
    // clocks, 8 times per work-group (stamps: loop 75 k clocks with nothing else on the SIMD against 37 k of MFMA issue).
    //   k = 0: the outer patch columns from the neighbour lanes;  k = 1 .. 6: Bt over columns 0, 5, 1, 2, 3, 4 (in place);
    //   k = 7 .. 12: Bt over row k - 7, written to the ring as soon as a position quad is complete.
    // Same operations in the same order per value as transform_put: bit-identical.
    // (Round 5: the same slices cut finer -- four parts of ~3 instructions, one behind EVERY MFMA of a quad instead of six behind two of
    // them -- are slower: one image at a time 163.4 against 169.8 Mpix/s, four in flight 269.2 against 272.8.  An instruction behind an
    // MFMA costs the issue slot between two MFMAs whatever its length; fewer, fuller gaps win.  Not kept.)
    float su[6][6], sp = 0.f, sq = 0.f, sr = 0.f, se = 0.f, sa = 0.f, sh4 = 0.f, sh5 = 0.f;
    auto bt_first = [&](float d0, float d1, float d2, float d3, float d4) __attribute__((always_inline)) {
        sp = fmaf(-4.f, d2, d4); sq = fmaf(-4.f, d1, d3); sr = d4 - d2; se = d3 - d1;
        sa = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    };
    auto bt_second = [&](float d1, float d3, float d5, float& t0, float& t1, float& t2, float& t3, float& t4, float& t5) __attribute__((always_inline)) {
        const float a5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        t0 = sa; t1 = sp + sq; t2 = sp - sq; t3 = fmaf(2.f, se, sr); t4 = fmaf(-2.f, se, sr); t5 = a5;
    };
    auto slice = [&](int k, int part, int half) __attribute__((always_inline)) {
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (part == 0) {
                    su[i][0] = dpp_from_left(pe[i], pr[i][3]);
                    if (SEG2) su[i][0] = mid_l ? pe[i] : su[i][0];
                } else {
                    su[i][5] = dpp_from_right(pe[i], pr[i][0]);
                    if (SEG2) su[i][5] = mid_r ? pe[i] : su[i][5];
                    su[i][1] = pr[i][0]; su[i][2] = pr[i][1]; su[i][3] = pr[i][2]; su[i][4] = pr[i][3];
                }
            }
        } else if (k <= 6) {
            const int c = k == 1 ? 0 : (k == 2 ? 5 : k - 2);
            if (part == 0) bt_first(su[0][c], su[1][c], su[2][c], su[3][c], su[4][c]);
            else bt_second(su[1][c], su[3][c], su[5][c], su[0][c], su[1][c], su[2][c], su[3][c], su[4][c], su[5][c]);
        } else {
            const int x = k - 7;
            if (part == 0) bt_first(su[x][0], su[x][1], su[x][2], su[x][3], su[x][4]);
            else {
                float v[6];
                bt_second(su[x][1], su[x][3], su[x][5], v[0], v[1], v[2], v[3], v[4], v[5]);
                f32x4* dst = &ring[((half * 4 + pw) * W4_QUADS + 3 * (x / 2)) * 64 + lane];
                if ((x & 1) == 0) { dst[0] = f32x4{v[0], v[1], v[2], v[3]}; sh4 = v[4]; sh5 = v[5]; }
                else { dst[64] = f32x4{sh4, sh5, v[0], v[1]}; dst[128] = f32x4{v[2], v[3], v[4], v[5]}; }
            }
        }
    };
    // WG8: a wave transforms in every OTHER iteration, behind scalar branches.  The compiler cannot see that the conditions of the
    // branches are one and the same, so every conditionally written value stays alive into the next conditional block -- across the
    // iterations that do not transform -- and the allocator spills 116-163 registers.  An empty asm that "defines" them in front of the
    // first conditional write ends the old value there: the live ranges are those of the 4-wave form again.  No instruction is emitted.
    auto kill_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { asm volatile("" : "=v"(pr[i])); asm volatile("" : "=v"(pe[i])); }
    };
    auto kill_slices = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jx = 0; jx < 6; ++jx) asm volatile("" : "=v"(su[i][jx]));
        asm volatile("" : "=v"(sp)); asm volatile("" : "=v"(sq)); asm volatile("" : "=v"(sr)); asm volatile("" : "=v"(se));
        asm volatile("" : "=v"(sa)); asm volatile("" : "=v"(sh4)); asm volatile("" : "=v"(sh5));
    };
    // WG8 without a barrier in the loop (W4_WG8_FLAGS): the two hazards of the ring as running counters in LDS.
    //   RAW  the 4 producing waves of a half add 1 to w8_flag[half] behind their last ring write (LDS executes a wave's operations in
    //        order: the add lands behind the writes); a wave starts iteration j -- reads half j & 1 -- once w8_flag[j & 1] >= 4 ((j + 1) >> 1);
    //   WAR  every wave adds 1 to w8_flag[2 + half] behind its last read of the half; a producer writes half h in iteration j (for j + 1)
    //        once w8_flag[2 + h] >= 8 ((j + 1) >> 1): everybody has finished iteration j - 1.
    // A wave is then held only by the waves it really depends on, 8 and 25 quads after they got there (the slack the barriers B1 / B2
    // describe), instead of meeting all seven others at the end of every iteration -- under the barrier the stamps show the waves of the
    // 8-wave form parked 31 k clocks of their 110 k (4-wave form: 8 k of 100 k; profiles/r06_w4_wg8.md).
    // One lane signals (exec = 1 around a ds_add_u32); the poll is bounded (W4_WG8_SPINS).
    const unsigned w8_base = (unsigned)(size_t)w8_flag;
    auto flag_add = [&](int k) __attribute__((always_inline)) {
        asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(w8_base + 4u * k), "v"(1u) : "memory");
    };
    // (the poll is ONE asm statement: a C++ loop inside the unrolled quad loop keeps the compiler from unrolling it -- the register rings
    // become indexed arrays in scratch: tried)
    auto flag_wait = [&](int k, unsigned target) __attribute__((always_inline)) {
        unsigned v, sv, spins;
        asm volatile("s_mov_b32 %2, 0\n"
                     "1:\tds_read_b32 %0, %3\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %1, %0\n\t"
                     "s_add_u32 %2, %2, 1\n\t"
                     "s_cmp_ge_u32 %1, %4\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "s_cmp_lt_u32 %2, %5\n\t"
                     "s_cbranch_scc1 1b\n"
                     "2:"
                     : "=&v"(v), "=&s"(sv), "=&s"(spins) : "v"(w8_base + 4u * k), "s"(target), "i"(W4_WG8_SPINS) : "memory", "scc");
    };
    // filter fragments: quad index Q = ks * 9 + q of this wave's channel tile: 1 KB per quad
    f32x4 fa[W4_RA];
    auto load_filter = [&](int slot, int Q) __attribute__((always_inline)) {       // slot = Q % W4_RA, passed as a constant
        fa[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(fr, fo, (cot * KS * W4_QUADS + Q) * 1024, 0));
    };

    // Operands of the epilogue, requested while the last iteration still runs (W4_PRE): BN scale / shift of the lane's four channels
    // and residual 1 of its first TWO channels.  The epilogue then asks for channel r + 2 when channel r is stored: with one
    // channel ahead (round 4's first version) every channel waited a full memory latency, 4 x ~5 k clocks at full load, stamps:
    // epilogue 20 k clocks of a wave's 117 k.  An absent residual is a zero-record descriptor: the load returns 0, no memory access.
    // The lane geometry is derived from the hardware lane id each time (kept alive across the loop it is spilled).
    const int img_bytes = COUT * HW * 4;
    f32x4 e1a[4], e1b[4];
    float scv[4], shv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { e1a[i] = e1b[i] = f32x4{0.f, 0.f, 0.f, 0.f}; scv[i] = shv[i] = 0.f; }
    auto epilogue_lanes = [&](unsigned* lo, int& kq_e) __attribute__((always_inline)) {
        int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_e));
        kq_e = lane_e >> 4;
        const int tx_e = SEG2 ? 8 * sx + (lane_e & 7) : 16 * sx + (lane_e & 15);
        const bool col_ok_e = 4 * tx_e < W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oy = 4 * (SEG2 ? 2 * ty + ((lane_e >> 3) & 1) : ty) + i;
            lo[i] = (col_ok_e && oy < H) ? (unsigned)((4 * kq_e * HW + oy * W + 4 * tx_e) * 4) : WN_OOB;
        }
    };
    auto request_first = [&](int first, int count) __attribute__((always_inline)) {       // residual 1 of channels first .. first + count - 1 (of 0, 1)
        unsigned lo[4];
        int kq_e;
        epilogue_lanes(lo, kq_e);
        if (first == 0 && !STATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = SHUF ? 4 * cot + kq_e : 16 * cot + 4 * kq_e + r;        // SHUF: one real channel per lane, four phases
                scv[r] = a.scale[ch]; shv[r] = a.shift[ch];
            }
        }
        if (RES > 0) {
            const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res1 ? a.res1 + (size_t)n * COUT * HW : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (first == 0) e1a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1r, lo[i], 16 * cot * HW * 4, 0));
                if (first + count > 1) e1b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1r, lo[i], (16 * cot + 1) * HW * 4, 0));
            }
        }
    };

    // ---- prologue: the input of iteration 0, the first filter fragments ----
    // (what a persistent work-group that inherits its first k-steps from its predecessor would save was measured with this block
    // compiled out -- wrong results, right timing: 8 Kodak maps 177.0 -> 170.8 us, a 4K map 434 -> 418 us, the bench step +1.5 %:
    // the other wave of the SIMD fills most of it.  Not built.)
    if (!WG8 || grp == 0) load_patch(pw);
#pragma unroll
    for (int Q = 0; Q < W4_RA - 1; ++Q) load_filter(Q, Q);
    // The NEXT layer's fragments into this XCD's L2 while this layer computes (one call at a time only: a lone launch starts every
    // layer on 2.36 MB of fragments no L2 has seen -- 64 layers x 2.36 MB -- and its one wave per SIMD waits for them; with several
    // images in flight the other launches cover the misses and the extra traffic costs more than it brings: 261.8 -> 257.3 Mpix/s,
    // round 4).  Work-groups are dealt round-robin to the XCDs: work-group j of an XCD's n touches one dword of every 128-byte line
    // of its 1 / n of the blob -- 3 loads per lane on a Kodak map.  LDS-DMA into a sink: no register is held for data nobody reads,
    // and the hidden loads can only make the compiler's vmcnt waits stricter (loads return in order), never weaker.
    if (a.wp_next) {
        constexpr unsigned LINES = 36u * CIN * COUT * 4u / 128u;
        const unsigned n_x = (gridDim.x + 7u) >> 3, j_x = blockIdx.x >> 3;
        const unsigned per = (LINES + n_x - 1) / n_x, l0 = j_x * per;
        const __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp_next, 0, 36 * CIN * COUT * 4, 0x00020000);
        const unsigned sink = (unsigned)(size_t)pf_sink;          // LDS byte address (wave-uniform)
        for (unsigned l = threadIdx.x; l < per; l += 64 * WAVES) {
            const unsigned off = (l0 + l) < LINES ? (l0 + l) * 128u : WN_OOB;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(nr), "s"(sink) : "memory");
        }
    }
    if (!WG8 || grp == 0) transform_put(0);
    if (WG8 && W4_WG8_FLAGS && threadIdx.x < 4) w8_flag[threadIdx.x] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#ifdef W4_STAMPS
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif
    // One iteration = 4 k-steps = 36 quads of 4 MFMAs.  Per quad, IN THIS ORDER: its 4 MFMAs, then the request of the filter
    // fragments W4_RA - 1 quads ahead into the ring slot the PREVIOUS quad consumed, then the B operands W4_RB - 1 quads ahead into the
    // slot the previous quad consumed: a register is overwritten by a load issued at least 4 MFMAs after its last reader.
    // `last`: the eighth iteration has no next k-steps to prepare; the epilogue's first operands are requested in its place.
    // Barriers (W4_SOFT, round 5): the ring's two hazards have 11 and 13 quads of slack, so neither needs the pipeline to stop.
    //   RAW  the other half is complete once every wave is past its last slice (quad 23): barrier B1 behind quad W4_B1 = 29; the B
    //        operands of the next iteration's first quads are then requested from quad 34 on, like any others, W4_RB - 1 ahead;
    //   WAR  the other half may be overwritten (first ring write of an iteration: quad 18; 11 with W4_SPREAD 0) once every wave has
    //        consumed its last operands of the previous iteration: barrier B2 behind quad W4_B2 = 8 of iterations 1 .. IT - 2.
    // Both sit between MFMA quads whose operands are already in registers, where the form of rounds 4 (W4_SOFT 0) ends every iteration
    // with `s_waitcnt lgkmcnt(0); s_barrier` and re-starts the B-operand reads behind it.  Built, tested (same values) and measured in
    // round 5 -- and it changes nothing: bench 282.3 / 281.8 / 281.3 against 283.0 / 281.9 / 281.6 Mpix/s, one image at a time 170.0
    // against 170.1, 8 Kodak maps 179.0 against 179.0 us.  The barrier is not where a wave's loop loses its 53 k - 37 k clocks; a lone
    // wave issues in order, and the dependent chains of its transform slices (6 vector instructions behind an MFMA, 4-8 clocks each)
    // take about the MFMA's own 32.  Kept as a switch, off.
    f32x4 bq[W4_RB];
#pragma unroll
    for (int q0 = 0; q0 < W4_RB - 1; ++q0) bq[q0] = ring_rd(0, q0);
    // `produce`: this wave prepares its k-step of the next iteration in this one (always in the 4-wave form; WG8: the group of the next half)
    auto iteration = [&](const int j, const int u2, const bool last, const bool produce) __attribute__((always_inline)) {      // reads ring half u2
        if (WG8 && W4_WG8_FLAGS && j > 0) flag_wait(u2, 4u * ((unsigned)(j + 1) >> 1));       // RAW: this half is complete
        if (!W4_SOFT && j > 0) {
#pragma unroll
            for (int q0 = 0; q0 < W4_RB - 1; ++q0) bq[q0] = ring_rd(u2, q0);
        }
#pragma unroll
        for (int lq = 0; lq < 36; ++lq) {                         // quad of the iteration: k-step lq / 9, positions 4 (lq % 9) ..
            const int q = lq % W4_QUADS;
            const int Q = j * 36 + lq;
            const int sk = lq - (W4_TURN + W4_GAP);               // slice of the spread turn that rides on this quad
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (4 * q + i < W4_ACC_A) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[4 * q + i]) : "v"(fa[lq % W4_RA][i]), "v"(bq[lq % W4_RB][i]));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[4 * q + i]) : "v"(fa[lq % W4_RA][i]), "v"(bq[lq % W4_RB][i]));
                if (WG8 && !last && sk == 0 && i == 0) kill_slices();       // (see kill_slices)
                if (W4_SPREAD && produce && !(W4_ABL & 4) && i < (W4_SLICE1 ? 1 : 2) && sk >= 0 && sk < 13) {      // (W4_ABL & 4: no transform at all)
                    __builtin_amdgcn_sched_barrier(0);
                    if (WG8 && W4_WG8_FLAGS && sk == 7 && i == 0) flag_wait(2 + (u2 ^ 1), 8u * ((unsigned)(j + 1) >> 1));   // WAR: before the first ring write
                    slice(sk, i, u2 ^ 1);
                    if (W4_SLICE1) slice(sk, 1, u2 ^ 1);
                    if (WG8 && W4_WG8_FLAGS && sk == 12 && i == (W4_SLICE1 ? 0 : 1)) flag_add(u2 ^ 1);                        // RAW: behind the last ring write
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if W4_ABL & 1
            if ((lq & 1) == 0)      // TIMING-ONLY ablation (wrong values): half of the filter-fragment requests
#endif
            if (!last || lq + W4_RA - 1 < 36) load_filter((lq + W4_RA - 1) % W4_RA, Q + W4_RA - 1);      // 36 % W4_RA == 0: the slot depends on lq only
            {
                const int ah = lq + W4_RB - 1;                    // the quad whose B operands are requested now
                if (ah < 36 && !((W4_ABL & 2) && (lq & 1)))   // (W4_ABL & 2: timing-only ablation, half of the B-operand reads)
                    bq[ah % W4_RB] = ring_rd(u2, ah);
                else if (W4_SOFT && !last) bq[ah % W4_RB] = ring_rd(u2 ^ 1, ah - 36);   // behind B1
            }
            // the wave's own k-step of the NEXT iteration: requested, then from W4_GAP quads later on transformed and written into
            // the other half (W4_SPREAD: in 13 slices behind MFMAs; else as one block)
            if (WG8 && !last && lq == W4_TURN) kill_patch();
            if (lq == W4_TURN && produce) load_patch(4 * (j + 1) + pw);
            if (!W4_SPREAD && lq == W4_TURN + W4_GAP && produce) transform_put(u2 ^ 1);
            if (last && lq == W4_PRE) request_first(0, W4_PRE_N);
            if (W4_SOFT && !last && lq == W4_B1) {                // the other half is complete (own ring writes landed: LDS works in order)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (W4_SOFT && !last && lq == W4_B2 && j > 0) __builtin_amdgcn_s_barrier();      // the other half has been read by everybody
            if (WG8 && W4_WG8_FLAGS && !last && lq == 36 - W4_RB + 1) flag_add(2 + u2);      // WAR: behind this wave's last read of the half (quad 35's operands: requested at 36 - W4_RB)
            __builtin_amdgcn_sched_barrier(0);                    // quads stay in program order: the rings are sized for exactly that
        }
        if (!W4_SOFT && !(WG8 && W4_WG8_FLAGS)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // next half complete, this half read by everybody
        }
    };
    // WG8: the group of the NEXT ring half produces in an iteration -- a scalar branch around each of its 13 slices and its patch request
    // (two copies of the loop, one per group, made the allocator spill 174 registers: tried)
    for (int jj = 0; jj < IT - 2; jj += 2) {
        iteration(jj, 0, false, !WG8 || grp == 1);
        iteration(jj + 1, 1, false, !WG8 || grp == 0);
    }
    iteration(IT - 2, 0, false, !WG8 || grp == 1);
    iteration(IT - 1, 1, true, false);
#ifdef W4_STAMPS
    const unsigned long long t_loop1 = __builtin_amdgcn_s_memtime();
#endif

    // inline asm is opaque to the hazard recogniser: pad the last MFMAs' latency, then pass every accumulator through an empty
    // volatile asm so that no read of it can be scheduled above the pad (conv3x3_wino_tn.hip found that the hard way)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int p = 0; p < 36; ++p) {
        if (p < W4_ACC_A) asm volatile("" : "+a"(acc[p]));
        else asm volatile("" : "+v"(acc[p]));
    }

    // ---- At M A, BN fold, activation, residuals, store: 4 channels x (4 x 4 pixels) per lane ----
    if (W4_PRE < 0) request_first(0, 2);
    else if (W4_PRE_N < 2) request_first(1, 1);
    unsigned lo[4];
    int kq_e;
    epilogue_lanes(lo, kq_e);
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    // At M A of "channel" r of the lane: (fenced step by step: left alone, the scheduler copies most of the 144 accumulators into
    // vector registers first and the operands requested early no longer fit; re-fenced at every use: the copy out of the
    // accumulator file stays where it is used)
    auto out_tile = [&](int r, float (&y)[4][4]) __attribute__((always_inline)) {
        float w_[6][4];                                           // columns transformed: w_[xi][j]
#pragma unroll
        for (int x = 0; x < 6; ++x) {
#pragma unroll
            for (int v = 0; v < 6; ++v) {
                if (6 * x + v < W4_ACC_A) asm volatile("" : "+a"(acc[6 * x + v]));
                else asm volatile("" : "+v"(acc[6 * x + v]));
            }
            w4_at(acc[6 * x][r], acc[6 * x + 1][r], acc[6 * x + 2][r], acc[6 * x + 3][r], acc[6 * x + 4][r], acc[6 * x + 5][r],
                  w_[x][0], w_[x][1], w_[x][2], w_[x][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            w4_at(w_[0][jx], w_[1][jx], w_[2][jx], w_[3][jx], w_[4][jx], w_[5][jx], y[0][jx], y[1][jx], y[2][jx], y[3][jx]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (SHUF) {
        // the lane's four "channels" are the phases (py, px) of ONE channel c = 4 cot + kq of the 2 H x 2 W output: rows 2 (4 ty + i) + py,
        // columns 8 tx + 2 j + px -- the two px phases of a row interleave into 8 consecutive pixels = two 16-byte stores
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)n * COUT * HW), 0, img_bytes, 0x00020000);
        int lane_s = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_s));
        const int tx_s = SEG2 ? 8 * sx + (lane_s & 7) : 16 * sx + (lane_s & 15);
        const int ty_s = SEG2 ? 2 * ty + ((lane_s >> 3) & 1) : ty;
        const bool col_ok_s = 4 * tx_s < W;
        const int so = 4 * cot * 4 * HW * 4;                      // channel 4 cot (+ kq: in the lane offset), 4 HW pixels per channel
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            f32x4 o0[4], o1[4];                                   // phase px = 0 / 1 of the row phase py, activated
            {
                float y[4][4];
                out_tile(2 * py, y);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) o0[i][jx] = fmaxf(fmaf(y[i][jx], scv[0], shv[0]), relu_lo);
                    asm volatile("" : "+v"(o0[i]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float y[4][4];
                out_tile(2 * py + 1, y);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) o1[i][jx] = fmaxf(fmaf(y[i][jx], scv[0], shv[0]), relu_lo);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int oy = 4 * ty_s + i;
                const unsigned lo_s = (col_ok_s && oy < H) ? (unsigned)(((lane_s >> 4) * 4 * HW + (2 * oy + py) * 2 * W + 8 * tx_s) * 4) : WN_OOB;
                const f32x4 oa = f32x4{o0[i][0], o1[i][0], o0[i][1], o1[i][1]}, ob = f32x4{o0[i][2], o1[i][2], o0[i][3], o1[i][3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oa), yr, lo_s, so, WT ? 16 : 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ob), yr, lo_s + 16u, so, WT ? 16 : 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (size_t)n * COUT * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res1 ? a.res1 + (size_t)n * COUT * HW : a.x), 0, a.res1 ? img_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2r = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res2 ? a.res2 + (size_t)n * COUT * HW : a.x), 0, a.res2 ? img_bytes : 0, 0x00020000);
    // Most layers have one residual or none (conv1 of a block has none, conv2 has the block input); every third block and the stack's
    // end add a second one: template parameter RES, so that the waits of the kernel without a second one count exactly the operations
    // in flight (a load that may or may not have been issued makes every wait behind it a wait for everything; two copies of the
    // epilogue inside one kernel spill).
    constexpr bool has1 = RES >= 1, has2 = RES >= 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 16 * cot + r;                              // + 4 kq: in the lane offset (4 kq HW)
        const int so = co * HW * 4;
        f32x4 e2[4];                                              // requested inside its channel's turn
#pragma unroll
        for (int i = 0; i < 4; ++i)
            e2[i] = has2 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r2r, lo[i], so, 0)) : f32x4{0.f, 0.f, 0.f, 0.f};
        float y[4][4];
        out_tile(r, y);
        f32x4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) o[i][jx] = STATS ? y[i][jx] : fmaxf(fmaf(y[i][jx], scv[r], shv[r]), relu_lo);
            if (has1) o[i] += (r & 1) ? e1b[i] : e1a[i];
            if (has2) o[i] += e2[i];
        }
        if (STATS) {
            // sum and sum of squares of the lane's 16 values that lie inside the map (rows beyond it carry lo[i] = WN_OOB), then over the 16
            // lanes of the row (= the 16 tiles of the segment, all of this channel): four DPP steps leave the total in the row's last lane.
            // Fixed order: bit-reproducible.
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = lo[i] != WN_OOB;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) { const float v = ok ? o[i][jx] : 0.f; s0 += v; s1 = fmaf(v, v, s1); }
            }
            // (row_shr:n with bound_ctrl: lanes without a source read 0)
#define W4_ROW_SHR_ADD(v, n) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + (n), 0xf, 0xf, true))
            W4_ROW_SHR_ADD(s0, 1); W4_ROW_SHR_ADD(s1, 1);
            W4_ROW_SHR_ADD(s0, 2); W4_ROW_SHR_ADD(s1, 2);
            W4_ROW_SHR_ADD(s0, 4); W4_ROW_SHR_ADD(s1, 4);
            W4_ROW_SHR_ADD(s0, 8); W4_ROW_SHR_ADD(s1, 8);
#undef W4_ROW_SHR_ADD
            int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if ((lane_t & 15) == 15) {
                float* d = a.stats + ((size_t)(co + 4 * (lane_t >> 4)) * a.ngroups + (seg - a.g0)) * 2;
                d[0] = s0; d[1] = s1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // residual 1 of channel r + 2 into the registers channel r just freed -- BEFORE this channel's stores: memory operations
        // complete in order, a load behind 16 stores waits for their acknowledgements
        if (has1 && r + 2 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1r, lo[i], so + 2 * HW * 4, 0));
                if (r & 1) e1b[i] = v; else e1a[i] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[i]), yr, lo[i], so, WT ? 16 : 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
#ifdef W4_LAUNCH_STAMPS
    // one record per WAVE, plain stores (atomics on one line from 768 waves cost 15 us per launch: tried): a.prof points at the launch's
    // first record; [0] first instruction, [1] all stores acknowledged -- both on the 100 MHz real-time counter every XCD shares --,
    // [2] shader clocks entry -> end (with [1] - [0]: the shader clock under this load), [3] MFMAs issued
    if (a.prof && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long ls_c1 = __builtin_amdgcn_s_memtime(), ls_r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = a.prof + 4 * ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6));
        d[0] = ls_r0; d[1] = ls_r1; d[2] = ls_c1 - ls_c0; d[3] = (unsigned long long)(36 * KS);
    }
#endif
#ifdef W4_STAMPS
    if (a.prof && (threadIdx.x & 63) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = a.prof + 6 * ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6));
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        d[0] = t_loop0 - t_entry; d[1] = t_loop1 - t_loop0; d[2] = t_end - t_loop1; d[3] = t_entry; d[4] = t_end;
        // HW_ID (wave slot, SIMD, CU, SE) and XCC_ID: which waves followed each other on the same slot (tools/w4prof.py: turnover)
        d[5] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32);
    }
#endif
}

#ifdef W4_STAMPS
// stamp builds only (never in the shipped library: tests/test_cpu_host.py checks the export list)
static void* g_w4_dbg = nullptr;
extern "C" void ic_wino4_debug_set_buffer(void* p) { g_w4_dbg = p; }
#endif
#ifdef W4_LAUNCH_STAMPS
// launch-stamp builds only (tools/w4_inflight_stamps.py): a device buffer of `capacity_waves` records of 4 x u64, zeroed by the caller.
// Launch i of the process (any stream, any instantiation) takes the next 4 x work-groups records; the host table gets
// (first record, waves, CIN * 1000 + COUT) per launch.
#include <atomic>
static unsigned long long* g_w4_ls = nullptr;
static long long g_w4_ls_cap = 0, g_w4_ls_max_launches = 0;
static std::atomic<long long> g_w4_ls_next{0}, g_w4_ls_launch{0};
static long long* g_w4_ls_table = nullptr;          // host array [max_launches][3]
extern "C" void ic_wino4_launch_stamps_set(void* dev_records, long long capacity_waves, long long* host_table, long long max_launches) {
    g_w4_ls = (unsigned long long*)dev_records; g_w4_ls_cap = capacity_waves; g_w4_ls_table = host_table; g_w4_ls_max_launches = max_launches;
    g_w4_ls_next = 0; g_w4_ls_launch = 0;
}
extern "C" long long ic_wino4_launch_stamps_count(void) { return g_w4_ls_launch.load(); }
#endif

extern "C" int ic_wino4_3x3_c128_supported(int N, int H, int W) {
    return N > 0 && H > 0 && W > 0 && (W & 3) == 0 && (long long)WN_C * H * W * 4 < (1ll << 31);
}

// segments of 16 tiles: 1 x 16 (wide maps) or 2 x 8 -- whichever covers the map with fewer of them (a tie keeps 1 x 16)
static inline long long w4_segments(int H, int W, bool seg2) {
    const long long th = ic_cdiv(H, 4), tw = ic_cdiv(W, 4);
    return seg2 ? ((th + 1) / 2) * ((tw + 7) / 8) : th * ((tw + 15) / 16);
}
static inline bool w4_seg2(int H, int W) { return w4_segments(H, W, true) < w4_segments(H, W, false); }

// work-groups of a launch: segments x 2 channel halves, two per CU
extern "C" long long ic_wino4_3x3_c128_workgroups(int N, int H, int W) {
    if (!ic_wino4_3x3_c128_supported(N, H, W)) return 0;
    return 2ll * N * w4_segments(H, W, w4_seg2(H, W));
}

// waves per work-group of the residual layers' launch: 8 (one work-group per segment, all 128 channels) where the caller asks for it
// (IC_CONV3_WINO4_WG8) or the launch has >= 2048 four-wave work-groups (a 4K map: +1 - 1.5 %, profiles/r06_w4_wg8.md), 4 otherwise
// and always with IC_CONV3_WINO4_WG4 or on maps with 2 x 8-tile segments
extern "C" int ic_wino4_3x3_c128_waves(int N, int H, int W, int flags) {
    if (!ic_wino4_3x3_c128_supported(N, H, W)) return 0;
    if ((flags & IC_CONV3_WINO4_WG4) || w4_seg2(H, W)) return 4;
    if (flags & IC_CONV3_WINO4_WG8) return 8;
    return ic_wino4_3x3_c128_workgroups(N, H, W) >= 2048 ? 8 : 4;
}

template <int RES, int CIN, int COUT, bool SHUF, bool SEG2, bool WG8>
static int w4_launch2(const float* x, const float* w_packed, const float* scale, const float* shift, const float* res1, const float* res2,
                      float* y, int N, int H, int W, int relu, int flags, hipStream_t st, const float* w_next = nullptr) {
    WnArgs a{};
    a.x = x; a.wp = w_packed; a.wp_next = w_next; a.scale = scale; a.shift = shift; a.res1 = res1; a.res2 = res2; a.y = y;
    a.N = N; a.H = H; a.W = W; a.relu = relu;
    a.grows = SEG2 ? ic_cdiv(ic_cdiv(H, 4), 2) : ic_cdiv(H, 4);
    a.gcols = SEG2 ? ic_cdiv(W, 32) : ic_cdiv(W, 64);
    a.xcd_runs = (flags & IC_CONV3_NO_XCD_RUNS) ? 0 : 1;
    a.g0 = 0; a.ngroups = N * a.grows * a.gcols;
#ifdef W4_STAMPS
    a.prof = (unsigned long long*)g_w4_dbg;
#endif
    constexpr int WAVES = WG8 ? 8 : 4;
    const long long wgs = (long long)(COUT / (16 * WAVES)) * a.ngroups;
#ifdef W4_LAUNCH_STAMPS
    if (g_w4_ls) {
        const long long first = g_w4_ls_next.fetch_add(WAVES * wgs), l = g_w4_ls_launch.load();
        if (first + WAVES * wgs <= g_w4_ls_cap && l < g_w4_ls_max_launches) {
            g_w4_ls_launch.fetch_add(1);
            a.prof = g_w4_ls + 4 * first;
            g_w4_ls_table[3 * l] = first; g_w4_ls_table[3 * l + 1] = WAVES * wgs; g_w4_ls_table[3 * l + 2] = CIN * 1000 + COUT;
        }
    }
#endif
    const dim3 grid((unsigned)wgs), block(64 * WAVES);
    if (wgs * WAVES <= W4_WT_MAX * 4) hipLaunchKernelGGL((wino4_3x3_kernel<true, RES, CIN, COUT, SHUF, SEG2, WG8>), grid, block, 0, st, a);     // a single round: write-through stores
    else hipLaunchKernelGGL((wino4_3x3_kernel<false, RES, CIN, COUT, SHUF, SEG2, WG8>), grid, block, 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
template <int RES, int CIN, int COUT, bool SHUF>
static int w4_launch(const float* x, const float* w_packed, const float* scale, const float* shift, const float* res1, const float* res2,
                     float* y, int N, int H, int W, int relu, int flags, hipStream_t st, const float* w_next = nullptr) {
    // 128-channel work-groups: the residual layers on maps with 1 x 16-tile segments only (the 2 x 8 segments of narrow maps carry per-lane
    // row conditions: with the counters' registers on top the allocator spills two of them -- isa_audit.py rule (S) -- and a map that narrow
    // has too few segments for one-work-group-per-CU launches anyway)
    if (CIN == 128 && COUT == 128 && ic_wino4_3x3_c128_waves(N, H, W, flags) == 8) {
        constexpr bool r = CIN == 128 && COUT == 128;                      // (keeps the other shapes from instantiating the form)
        return w4_launch2<RES, CIN, COUT, SHUF, false, r>(x, w_packed, scale, shift, res1, res2, y, N, H, W, relu, flags, st, w_next);
    }
    if (w4_seg2(H, W)) return w4_launch2<RES, CIN, COUT, SHUF, true, false>(x, w_packed, scale, shift, res1, res2, y, N, H, W, relu, flags, st, w_next);
    return w4_launch2<RES, CIN, COUT, SHUF, false, false>(x, w_packed, scale, shift, res1, res2, y, N, H, W, relu, flags, st, w_next);
}

int icx_wino4_3x3_c128_next(const float* x, const float* w_packed, const float* scale, const float* shift, const float* res1, const float* res2,
                            float* y, int N, int H, int W, int relu, int flags, const float* w_packed_next, hipStream_t st) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y);
    IC_CHECK_ARG(N > 0 && H > 0 && W > 0);
    if (!ic_wino4_3x3_c128_supported(N, H, W)) return IC_ERR_UNSUPPORTED;
    if (res2) return w4_launch<2, 128, 128, false>(x, w_packed, scale, shift, res1, res2, y, N, H, W, relu, flags, st, w_packed_next);
    return w4_launch<1, 128, 128, false>(x, w_packed, scale, shift, res1, nullptr, y, N, H, W, relu, flags, st, w_packed_next);
}
// ---- the training step's forward convolution: raw output + per-segment channel sums for BatchNorm (STATS instantiations) ----
extern "C" long long ic_wino4_3x3_c128_stats_parts(int N, int H, int W) {
    if (!ic_wino4_3x3_c128_supported(N, H, W)) return 0;
    return (long long)N * w4_segments(H, W, w4_seg2(H, W));
}
template <bool SEG2>
static int w4_launch_stats(const float* x, const float* w_packed, float* y, float* stats, int N, int H, int W, int flags, hipStream_t st) {
    WnArgs a{};
    a.x = x; a.wp = w_packed; a.y = y; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.relu = 0;
    a.grows = SEG2 ? ic_cdiv(ic_cdiv(H, 4), 2) : ic_cdiv(H, 4);
    a.gcols = SEG2 ? ic_cdiv(W, 32) : ic_cdiv(W, 64);
    a.xcd_runs = (flags & IC_CONV3_NO_XCD_RUNS) ? 0 : 1;
    a.g0 = 0; a.ngroups = N * a.grows * a.gcols;
    const long long wgs = 2ll * a.ngroups;
    const dim3 grid((unsigned)wgs), block(256);
    if (wgs <= W4_WT_MAX) hipLaunchKernelGGL((wino4_3x3_kernel<true, 0, 128, 128, false, SEG2, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wino4_3x3_kernel<false, 0, 128, 128, false, SEG2, false, true>), grid, block, 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
extern "C" int ic_wino4_3x3_c128_raw_stats_f32(const float* x, const float* w_packed, float* y, float* stats, int N, int H, int W,
                                               int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && y && stats && N > 0 && H > 0 && W > 0);
    if (!ic_wino4_3x3_c128_supported(N, H, W)) return IC_ERR_UNSUPPORTED;
    if (w4_seg2(H, W)) return w4_launch_stats<true>(x, w_packed, y, stats, N, H, W, flags, (hipStream_t)stream);
    return w4_launch_stats<false>(x, w_packed, y, stats, N, H, W, flags, (hipStream_t)stream);
}

extern "C" int ic_wino4_3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                            const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                                            int flags, ic_stream_t stream) {
    return icx_wino4_3x3_c128_next(x, w_packed, scale, shift, res1, res2, y, N, H, W, relu, flags, nullptr, (hipStream_t)stream);
}

// ---- the two large 5x5 / stride-2 layers as F(4x4) 3x3 convolutions over phases (packing modes 2 / 3 above) -------------------------
// h2 (autoencoder.py:223): x_phases [N][4 x 64][H][W] = the four phases of the 64-channel 2H x 2W input (channel (2 py + px) * 64 + c
// holds X[c][2 i + py][2 j + px]; h1 writes it that way on request, ic_space_to_depth2_f32 makes it from a plain tensor) -> y [N][128][H][W].
// h12 (autoencoder.py:264): x [N][128][H][W] -> y [N][64][2H][2W], plain layouts.  W % 4 == 0; 10.07 GFLOP direct -> 3.62 executed.
extern "C" int ic_wino4_conv5s2_supported(int N, int H, int W) {
    return N > 0 && H > 0 && W > 0 && (W & 3) == 0 && (long long)256 * H * W * 4 < (1ll << 31);
}
extern "C" long long ic_wino4_conv5s2_workgroups(int N, int H, int W, int transposed) {
    if (!ic_wino4_conv5s2_supported(N, H, W)) return 0;
    return (transposed ? 4ll : 2ll) * N * w4_segments(H, W, w4_seg2(H, W));
}
extern "C" int ic_wino4_conv5s2_c64_c128_bn_act_f32(const float* x_phases, const float* w_packed, const float* scale, const float* shift,
                                                    float* y, int N, int H, int W, int relu, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x_phases && w_packed && scale && shift && y);
    if (!ic_wino4_conv5s2_supported(N, H, W)) return IC_ERR_UNSUPPORTED;
    return w4_launch<0, 256, 128, false>(x_phases, w_packed, scale, shift, nullptr, nullptr, y, N, H, W, relu, flags, (hipStream_t)stream);
}
extern "C" int ic_wino4_deconv5s2_c128_c64_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                                      float* y, int N, int H, int W, int relu, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && w_packed && scale && shift && y);
    if (!ic_wino4_conv5s2_supported(N, H, W)) return IC_ERR_UNSUPPORTED;
    return w4_launch<0, 128, 256, true>(x, w_packed, scale, shift, nullptr, nullptr, y, N, H, W, relu, flags, (hipStream_t)stream);
}

// [N][C][2H][2W] -> [N][4][C][H][W]: phase (py, px) of every channel as its own plane (what ic_wino4_conv5s2_c64_c128_bn_act_f32 reads)
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % W); long long t = i / W;
        const int h = (int)(t % H); t /= H;
        const int c = (int)(t % C); t /= C;
        const int ph = (int)(t & 3); const long long n = t >> 2;
        y[i] = x[((n * C + c) * 2 * H + 2 * h + (ph >> 1)) * 2 * W + 2 * w + (ph & 1)];
    }
}
extern "C" int ic_space_to_depth2_f32(const float* x, float* y, int N, int C, int H2, int W2, ic_stream_t stream) {
    IC_CHECK_ARG(x && y && N > 0 && C > 0 && H2 > 0 && W2 > 0 && !(H2 & 1) && !(W2 & 1));
    const long long total = (long long)N * C * H2 * W2;
    const long long g = (total + 255) / 256;
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, x, y, C, H2 / 2, W2 / 2, total);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
