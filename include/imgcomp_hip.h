/* imgcomp_hip.h -- C ABI of libimgcomp_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the data-parallel hot path of fab-jul/imgcomp-cvpr.  The reference has
 * no FFI of its own: its boundary is the Python plugin surface (autoencoder.get_network_cls,
 * probclass.get_network_cls, quantizer.quantize) whose ops TensorFlow dispatches to device
 * kernels.  Every entry point below replaces the TF op call sites it cites (file:line relative
 * to the reference's code/ directory); the Python mirror of the plugin surface
 * (imgcomp_cvpr_amd/{autoencoder,probclass,quantizer}.py) binds them with ctypes.
 *
 * Conventions
 *  - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 *    buffer including workspaces; the library never allocates device memory;
 *  - tensors are dense fp32 NCHW, symbols are int64 (reference: quantizer.py:46);
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous on it and re-entrant: the library keeps NO
 *    process-wide mutable state.  Everything that selects a kernel form, a launch plan or a test-only code path is a
 *    bit of the per-call `flags` argument of the entry point concerned (IC_CONV3_*, IC_EDGE_*, IC_PC_*; 0 = automatic),
 *    so two host threads may drive two pipelines through the library concurrently.  (Profiling builds of single
 *    translation units -- -DWN_PROF, -DH13_PROF, -DIC_TUNING -- add ic_*_debug_* setters; they are not part of this ABI
 *    and not compiled into libimgcomp_hip.so.);
 *  - return value: 0 = ok, < 0 = argument error (IC_ERR_*), > 0 = hipError_t of a failed launch;
 *  - weights use the reference's TF variable layouts unless a function says "packed".
 */
#ifndef IMGCOMP_HIP_H
#define IMGCOMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IC_ABI_VERSION 2

#define IC_OK 0
#define IC_ERR_ARG (-1)          /* null pointer / non-positive extent */
#define IC_ERR_UNSUPPORTED (-2)  /* shape or option outside what the kernels implement */
#define IC_ERR_WORKSPACE (-3)    /* workspace too small */

typedef void* ic_stream_t;

/* ---- per-call plan flags ------------------------------------------------------------------------------------------
 * 3x3 128->128 layer (ic_conv3x3_c128_bn_act_f32, ic_wino3x3_c128_bn_act_f32, ic_conv3x3_c128_auto_f32,
 * ic_wino3x3_c128_workgroups, ic_conv3x3_c128_pick_algo, and passed through by ic_ae_encode_f32 / ic_ae_decode_f32).
 * Bits 0-3: the form.  0 = automatic (from the shape: Winograd wherever 31-bit offsets reach, decomposition by the cost
 * model of conv3x3_wino.hip:wino_plan); the others force one form for the whole launch (tests, benchmarks). */
#define IC_CONV3_FORM_MASK        0x0f
#define IC_CONV3_AUTO             0x00
#define IC_CONV3_DIRECT           0x01   /* implicit-GEMM direct form (needs the `both` blob or the direct packing) */
#define IC_CONV3_WINO             0x02   /* Winograd, decomposition automatic */
#define IC_CONV3_WINO_WHOLEK      0x03   /* 32x32 jobs, 4 channel tiles per work-group, input transform shared through LDS */
#define IC_CONV3_WINO_WHOLEK_PW   0x04   /* the same with a per-wave input transform (the whole-K form of odd widths) */
#define IC_CONV3_WINO_KSPLIT      0x05   /* one channel tile per work-group, 4 K-quarters summed through LDS */
#define IC_CONV3_WINO_T16         0x06   /* 16x16 jobs, two-slot ring (round-1 kernel; even widths).  TUNING builds only, see below */
#define IC_CONV3_WINO_SEG1        0x07   /* 16 channels x NB segments of 16 tiles per wave, NB = 1 / 2 / 3 (even widths) */
#define IC_CONV3_WINO_SEG2        0x08
#define IC_CONV3_WINO_SEG3        0x09
#define IC_CONV3_WINO_PAIR        0x0a   /* tile group x 32 channels per work-group, waves split the 16 positions, two work-groups per CU
                                            (even widths; conv3x3_wino_tp.hip).  ic_wino3x3_c128_plan reports it as segment jobs with nb = -1.
                                            TUNING builds only */
#define IC_CONV3_WINO4            0x0b   /* Winograd F(4x4,3x3) (conv3x3_wino4.hip; W % 4 == 0, else the F(2x2) plan) */
#define IC_CONV3_NO_WINO4         0x800000   /* automatic plan: F(2x2) forms only (A/B runs, bit-identity tests between F(2x2) forms) */
#define IC_CONV3_WINO4_WG8        0x8000000  /* F(4x4) kernel: ONE 8-wave work-group per segment covers all 128 output channels (the input
                                              transform is made once per segment instead of once per 64-channel half); bit-identical to the
                                              4-wave form.  A/B runs and tests; the plan's own choice: ic_wino4_3x3_c128_waves */
#define IC_CONV3_WINO4_WG4        0x10000000 /* F(4x4) kernel: always the 4-wave form (two 64-channel work-groups per segment); A/B runs */
#define IC_CONV3_WINO4_BITS       (IC_CONV3_WINO4_WG8 | IC_CONV3_WINO4_WG4)
/* h2 / h12 of the whole-network entry points (ic_ae_encode_f32 / ic_ae_decode_f32): their filter blobs were packed by
 * ic_pack_conv5s2_both_f32 (MFMA fragments followed by the F(4x4)-over-phases fragments), so the library may run them on the
 * F(4x4) kernel where ic_conv3x3_c128_pick_form picks it for the residual stack of the same call or h2's launch has >= 160
 * work-groups (a Kodak map); _WINO4: wherever the shape allows; _NO_WINO4: never.  Without IC_CONV5_BOTH_PACKED the blobs are ic_pack_conv2d_mfma_f32's and the direct kernels run. */
#define IC_CONV5_BOTH_PACKED      0x1000000
#define IC_CONV5_WINO4            0x2000000
#define IC_CONV5_NO_WINO4         0x4000000
/* IC_CONV3_WINO_T16, IC_CONV3_WINO_PAIR and IC_CONV3_STACK_KERNEL name forms that were built, tested bit-identical and measured
 * slower than or level with what the plan picks.  The shipped library does not carry them (`make TUNING=1` does):
 * ic_build_has_tuning_forms() tells, forcing T16 / PAIR without them returns IC_ERR_UNSUPPORTED, IC_CONV3_STACK_KERNEL is ignored. */
int ic_build_has_tuning_forms(void);
/* partly filled rounds stay one-work-group-per-CU and the CUs beyond the tile groups stay free: a caller's independent
 * branch runs there on a CU-range stream (ic_stream_create_cu_range; imgcomp_cvpr_amd/streams.py) */
#define IC_CONV3_LEAVE_IDLE_CUS   0x10
/* whole-network calls (ic_ae_*): IC_CONV3_LEAVE_IDLE_CUS applies to the first n 3x3 launches of the call only (0 = all):
 * a side branch that needs its CUs for less than the whole residual stack */
#define IC_CONV3_LEAVE_IDLE_LAYERS(n) (((n) & 0x7f) << 12)
#define IC_CONV3_NO_XCD_RUNS      0x20   /* natural tile order instead of contiguous runs of tiles per XCD (A/B runs) */
#define IC_CONV3_STACK_KERNEL      0x80   /* whole-network / residual-stack calls: where the stack's NB-segment jobs fit the chip in one
                                            round, run its 6B+2 layers as ONE persistent launch (conv3x3_wino_stack.hip) instead of one
                                            launch per layer.  Bit-identical; measured level with per-layer launches, so
                                            it is an option, not the default */
#define IC_CONV3_PACKED_TRANSFORM 0x40   /* NB-segment kernels: input transform on v_pk_add_f32 instead of single adds (A/B) */
/* the caller keeps n (2..15) INDEPENDENT calls of this shape in flight on different streams (the images of an evaluation set,
 * val.py:157-158): a launch no longer has to fill the chip by itself, so the plan takes the form that costs the least CU-time --
 * 32 x 32 whole-K jobs, whose 192 work-groups for a Kodak map leave a quarter of the chip to the neighbouring stream's launch
 * instead of idle -- for launches of at least 128 tile groups (smaller maps keep the one-launch plan: their work-groups are short
 * and the overlap alone fills the chip).  n = 0 / 1: plan for one launch at a time. */
#define IC_CONV3_IN_FLIGHT(n)     (((n) & 0xf) << 19)
#define IC_CONV3_DIRECT_VARIANT(v) ((((v) + 1) & 0xf) << 8)   /* direct form: force tile variant v (0..9); tests */
/* h13 (ic_deconv2d_bn_act_f32, 5x5/2 transposed, 64 -> <= 4): tiles per work-group, 0 = automatic; tests */
#define IC_EDGE_TILES_PER_WG(n)   ((n) & 0xff)
/* ic_pc_decode_f32: always the launch-per-layer loop, also for k = 24 (tests) */
#define IC_PC_DECODE_PER_LAYER    0x01
/* ic_pc_decode_f32, k = 24: the persistent work-group that recomputes each symbol's whole 5x9x9 context (round-1 kernel)
 * instead of the one with activation caches (tests, A/B) */
#define IC_PC_DECODE_RECOMPUTE    0x02

int ic_abi_version(void);
/* static string for a return code of this library (hipGetErrorString for codes > 0) */
const char* ic_strerror(int code);

/* host utility: CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 to start) -- the checksum of TF-1
 * checkpoint blocks and tensors (saver.py:46-100 -> tf.train.Saver; imgcomp_cvpr_amd/tf_checkpoint.py). */
uint32_t ic_crc32c(const void* host_data, size_t bytes, uint32_t crc);

/* ---------------------------------------------------------------------------------------------
 * Generic direct convolution + folded BatchNorm + activation (+ up to two residual adds).
 * Replaces slim.conv2d(..., normalizer_fn=slim.batch_norm) at autoencoder.py:222 (h1), :223 (h2),
 * :237 (to_bn), :285 (residual 3x3 convs; the MFMA kernel below is the fast path for those).
 *   x      (N,Cin,H,W)          w  [KH,KW,Cin,Cout]  (TF conv2d filter layout)
 *   y      (N,Cout,OH,OW), OH = ceil(H/stride), TF 'SAME' padding (pad_before = total/2)
 *   y = act(conv(x') * scale[co] + shift[co]) + res1 + res2      (res1/res2 nullable, shape of y)
 *   x' = (x - in_mean[ci]) / in_std[ci] when in_mean != NULL (autoencoder.py:136-144, applied
 *        before the zero padding, as the reference normalises before the conv pads)
 *   scale = gamma / sqrt(moving_var + 1e-5), shift = beta - moving_mean * scale (autoencoder.py:114-125)
 *   relu: 0/1.  stride: 1 or 2.
 */
int ic_conv2d_bn_act_f32(const float* x, const float* w, const float* scale, const float* shift,
                         const float* res1, const float* res2, float* y,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int relu,
                         const float* in_mean, const float* in_std, ic_stream_t stream);

/* Stride-2 'SAME' transposed convolution + folded BN + activation (+ de-normalise + clip).
 * Replaces slim.conv2d_transpose at autoencoder.py:251 (from_bn), :264 (h12), :265 (h13) and
 * _denormalize/_clip_to_image_range (:146-158) when out_mean != NULL.
 *   x (N,Cin,H,W)   w [KH,KW,Cout,Cin] (TF conv2d_transpose filter layout)   y (N,Cout,2H,2W)
 *   y = act(deconv(x) * scale + shift);  if out_mean: y = clip(y * out_std[co] + out_mean[co], 0, 255)
 */
int ic_deconv2d_bn_act_f32(const float* x, const float* w, const float* scale, const float* shift,
                           float* y, int N, int Cin, int H, int W, int Cout, int KH, int KW, int relu,
                           const float* out_mean, const float* out_std, int flags, ic_stream_t stream);
/* The three edge layers of the autoencoder run on the matrix cores inside the two entry points above, straight
 * from the TF filter layouts: h1 (conv 5x5/2, 3 -> 64, :222), from_bn (deconv 3x3/2, C = 32|64 -> 128, :251) and
 * h13 (deconv 5x5/2, 64 -> <= 4, :265).  Every other shape takes the generic direct kernels.
 * flags: IC_EDGE_TILES_PER_WG(n) for the h13 kernel, 0 = automatic. */

/* ---------------------------------------------------------------------------------------------
 * MFMA fast path for the 64 residual 3x3 convs (128 -> 128 channels, stride 1):
 * autoencoder.py:274-287 called from :225-234 and :253-262 (~95 % of the path's FLOPs).
 * ic_pack_conv3x3_c128_f32 re-orders a TF-layout filter [3,3,128,128] into the MFMA A-fragment
 * order once (147,456 floats in, 147,456 floats out); ic_conv3x3_c128_bn_act_f32 consumes it.
 *   y = act(conv3x3(x) * scale + shift) + res1 + res2       x, y, res*: (N,128,H,W)
 */
size_t ic_conv3x3_c128_packed_floats(void);
int ic_pack_conv3x3_c128_f32(const float* w_tf, float* w_packed, ic_stream_t stream);
int ic_conv3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale,
                               const float* shift, const float* res1, const float* res2, float* y,
                               int N, int H, int W, int relu, int flags, ic_stream_t stream);

/* The same layer (3x3, stride 1, 128 -> 128, autoencoder.py:274-287) in Winograd F(2x2,3x3) form: 16/36 of the
 * multiply-adds of the direct form; input transforms in registers, shared between the waves of a work-group through LDS.
 * Same contract as ic_conv3x3_c128_bn_act_f32 with its own packed filter: G g Gt (16 x 128 x 128 floats) in MFMA
 * A-fragment order, stored twice -- for v_mfma_f32_32x32x2_f32 and for v_mfma_f32_16x16x4_f32 work-groups -- so
 * ic_wino3x3_c128_packed_floats() = 2 x 16 x 128 x 128; backward != 0 packs the adjoint filter used for the data
 * gradient in training.
 * Results differ from the direct form by fp32 rounding only (both within 1e-4 of the float64 oracle). */
size_t ic_wino3x3_c128_packed_floats(void);
int ic_pack_wino3x3_c128_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream);
/* the same for `layers` filters in one launch: w_tf_table_dev is a DEVICE array of `layers` device pointers, layer l is
 * packed to w_packed + l * ic_wino3x3_c128_packed_floats() (the training step re-packs all 64 filters twice per step) */
int ic_pack_wino3x3_c128_batch_f32(const float* const* w_tf_table_dev, float* w_packed, int layers, int backward,
                                   ic_stream_t stream);
int ic_wino3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale,
                               const float* shift, const float* res1, const float* res2, float* y,
                               int N, int H, int W, int relu, int flags, ic_stream_t stream);
/* Launch plan (conv3x3_wino.hip:wino_plan; flags = 0): the form is chosen per call from the number of tile groups
 * (4 x 32 output pixels each) by a cost model in matrix-pipe clocks -- full rounds of 256 whole-K work-groups, a small
 * remainder as K-split work-groups, or the whole map as NB-segment jobs (NB = 1, 2, 3) when that fills the chip more
 * evenly (a Kodak map: 384 segments = 256 work-groups of NB = 3, one wave-job per SIMD).
 * ic_wino3x3_c128_workgroups: CUs the launch(es) for this shape and these flags keep busy at a time (>= 256: the whole
 * chip, in rounds) -- what a caller sizing a CU-range stream for an independent branch needs to know. */
long long ic_wino3x3_c128_workgroups(int N, int H, int W, int flags);
/* the plan itself (host arithmetic only): tile groups given to each form,
 * plan_out = {whole-K, NB-segment jobs, NB of those jobs, 16x16 two-slot, K-split} */
int ic_wino3x3_c128_plan(int N, int H, int W, int flags, long long plan_out[5]);

/* Both forms behind ONE packed filter [direct fragments | Winograd fragments]; this is what ic_ae_encode_f32 /
 * ic_ae_decode_f32 expect in their tables for the 3x3 layers and what the training step uses (backward != 0: adjoint).
 * ic_conv3x3_c128_auto_f32 picks the form per launch from (N, H, W) and the caller's flags -- ic_conv3x3_c128_pick_algo
 * returns the choice (0 direct, 1 Winograd). */
size_t ic_conv3x3_c128_both_packed_floats(void);
int ic_pack_conv3x3_c128_both_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream);
int ic_conv3x3_c128_pick_algo(int N, int H, int W, int flags);
int ic_conv3x3_c128_auto_f32(const float* x, const float* w_both, const float* scale, const float* shift,
                             const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                             int flags, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MFMA path for the strided 5x5 layers around the residual stacks: h2 (autoencoder.py:223, conv 64->128),
 * to_bn (:237, conv 128->C+1) and h12 (:264, transposed conv 128->64).  Filters are re-ordered once by
 * ic_pack_conv2d_mfma_f32 (TF layout in: conv [5,5,cin,cout], transposed [5,5,cout,cin]); a transposed
 * conv is stored and executed as its four output phases.  Unsupported shapes return IC_ERR_UNSUPPORTED
 * (ic_conv2d_mfma_packed_floats returns 0) -- use ic_conv2d_bn_act_f32 / ic_deconv2d_bn_act_f32 then.
 *   y = act(conv(x) * scale + shift);  x (N,Cin,H,W) -> y (N,Cout,H/2,W/2) or, transposed, (N,Cout,2H,2W)
 */
size_t ic_conv2d_mfma_packed_floats(int KH, int KW, int Cin, int Cout, int stride, int transposed);
int ic_pack_conv2d_mfma_f32(const float* w_tf, float* w_packed, int KH, int KW, int Cin, int Cout,
                            int stride, int transposed, ic_stream_t stream);
int ic_conv2d_mfma_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                              float* y, int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride,
                              int transposed, int relu, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Importance map + quantiser.
 * ic_quantize_f32 replaces quantizer.quantize / _quantize1d (quantizer.py:37-100):
 *   d_j = (z - c_j)^2; qsoft = sum_j softmax(-sigma d)_j c_j; symbols = first argmax_j of
 *   (-1e7 * d_j); qhard = c[symbols].   z, qsoft, qhard: `count` floats; symbols: `count` int64.
 *   Any output pointer may be NULL.  L <= 16.
 * ic_heatmap_quantize_f32 fuses _get_heatmap3D + _mask_with_heatmap (autoencoder.py:171-200)
 * with the quantiser and the STE forward value qbar = qsoft + (qhard - qsoft) (:127-134):
 *   bottleneck (N,C+1,h,w) -> heatmap, z, qsoft, qhard, qbar (N,C,h,w) fp32, symbols int64.
 */
int ic_quantize_f32(const float* z, const float* centers, int L, float sigma,
                    float* qsoft, float* qhard, int64_t* symbols, long long count, ic_stream_t stream);
int ic_heatmap_quantize_f32(const float* bottleneck, const float* centers, int L, float sigma,
                            float* heatmap, float* z, float* qsoft, float* qhard, float* qbar,
                            int64_t* symbols, int N, int C, int h, int w, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Context model (res_shallow): probclass.py:63-106 (bitcost), :130-135 (logits), :214-221
 * (_ResShallow._logits), :227-261 (conv3d), :150-176 (masks), :268-292 (pad).
 * The volume is padded on load with pad_value (= centers[0], probclass.py:59-61): depth front 4,
 * H/W 4 each side; four masked VALID (2,3,3) conv3d layers 1->k->k->k->L with one residual;
 * the last layer keeps conv3d's default ReLU.  Logits for ALL positions are produced in parallel.
 *   q        (N,C,h,w) fp32     symbols (N,C,h,w) int64
 *   wtab[9]  device pointers {w0,b0,w1,b1,w2,b2,w3,b3, packed}; w* in TF layout [2,3,3,cin,cout], UNMASKED
 *            (the masks are applied by skipping the dead taps); layers: conv0, res1/conv1,
 *            res1/conv2, conv2(final).  packed: NULL, or the matrix-core fragment packing of w1..w3 made ONCE with
 *            ic_pc_pack_filters_f32 (ic_pc_packed_floats(k, L) floats; inference: the weights do not change between
 *            calls) -- with NULL every call packs into its workspace first (training: the weights change every step)
 *   logits   (N,C,h,w,L) fp32 (nullable for bitcost)     bits (N,C,h,w) fp32 = CE * log2(e)
 *   workspace: ic_pc_workspace_bytes(N,C,h,w,k) bytes.
 */
size_t ic_pc_workspace_bytes(int N, int C, int h, int w, int k);
size_t ic_pc_packed_floats(int k, int L);       /* 0: this (k, L) runs the any-shape VALU kernels, nothing to pack */
int ic_pc_pack_filters_f32(const float* const* wtab_host, int k, int L, float* packed, ic_stream_t stream);
int ic_pc_logits_f32(const float* q, const float* const* wtab_host, int k, int L, float pad_value,
                     float* logits, int N, int C, int h, int w,
                     void* workspace, size_t workspace_bytes, ic_stream_t stream);
/* same network on an ALREADY padded volume (N,D,H,W) -> logits (N,D-4,H-8,W-8,L); this is
 * _Network3D.logits (probclass.py:130-135) as PredictionNetwork calls it on a (5,9,9) context
 * (probclass.py:436-442).  workspace: ic_pc_workspace_bytes(N, D-4, H-8, W-8, k). */
int ic_pc_logits_padded_f32(const float* vol, const float* const* wtab_host, int k, int L,
                            float* logits, int N, int D, int H, int W,
                            void* workspace, size_t workspace_bytes, ic_stream_t stream);
int ic_pc_bitcost_f32(const float* q, const int64_t* symbols, const float* const* wtab_host, int k, int L,
                      float pad_value, float* logits, float* bits, int N, int C, int h, int w,
                      void* workspace, size_t workspace_bytes, ic_stream_t stream);
/* logits (count, L) -> freqs (count, L) int64 = max(int64(softmax(logits) * resolution), 1) and optionally the
 * probabilities pr (count, L): PredictionNetwork.freqs / get_freqs (probclass.py:443-444, :465-476).  The row
 * expression is fixed, so tables built from all-position logits equal tables built context by context. */
int ic_pc_logits_to_freqs_f32(const float* logits, long long count, int L, float resolution,
                              int64_t* freqs, float* pr, ic_stream_t stream);

/* Row N3 -- the decoder side of the real-bpp path without the host in the loop (bit_counter.py:137-164 does one
 * context -> table -> arithmetic-decoder step per host round trip).  Decodes the whole (C,h,w) symbol volume from an
 * arithmetic-coded bit stream: per symbol the same context-model kernels as ic_pc_logits_padded_f32 run on the
 * gathered 5x9x9 context (identical tables by construction), and one small kernel builds the integer table, steps the
 * 32-bit range decoder (arithmetic_coding.py restated for the device), stores the symbol and gathers the next
 * context.  k = 24 runs the whole volume as ONE persistent work-group (coder state in registers, activations in LDS);
 * other k enqueue five launches per symbol (volumes of >= 256 symbols replay a captured hipGraph of 128 symbol steps on a
 * private stream and wait for it before returning).
 *   bitstream: device bytes written by the encoder (symbols 1..n-1 in raster order C,H,W; bit_counter.py:103-131)
 *   first_sym: the uncoded first symbol;  centers: device [L];  resolution: 1e9 (probclass.py:443)
 *   symbols: out, device int64 (C,h,w);  status: out, device int (0 ok, 1 = a table's total exceeded the coder's range)
 *   wtab_host / k / L as ic_pc_logits_f32.  workspace: ic_pc_decode_workspace_bytes(C, h, w, k). */
size_t ic_pc_decode_workspace_bytes(int C, int h, int w, int k);
/* flags: 0 lets k = 24 run as ONE persistent work-group with activation caches (one new voxel per layer per symbol);
 * IC_PC_DECODE_RECOMPUTE: the persistent work-group without caches; IC_PC_DECODE_PER_LAYER: the launch-per-layer loop */
int ic_pc_decode_f32(const uint8_t* bitstream, long long nbytes, int first_sym, const float* const* wtab_host,
                     const float* centers, int k, int L, float resolution, int64_t* symbols, int* status,
                     int C, int h, int w, void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream);
/* bits -> sum(bits) (bits.py:4-14 numerator); deterministic two-stage reduction.
 * partial: >= 1024 floats of scratch.  out_sum: 1 float. */
int ic_sum_f32(const float* v, long long count, float* partial, float* out_sum, ic_stream_t stream);
/* sum(v) / denom with the same reduction (bits.bitcost_to_bpp: bits.py:4-14 whole); the fp32 division is IEEE */
int ic_mean_f32(const float* v, long long count, float denom, float* partial, float* out, ic_stream_t stream);
/* the same for `rows` consecutive rows of `count` values (the bpp of every image of a batch, val.py): out[r]; one `partial` for all rows */
int ic_mean_rows_f32(const float* v, int rows, long long count, float denom, float* partial, float* out, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-network entry points for the CVPR autoencoder (autoencoder.py:218-268): one host call
 * enqueues every stage, so batch-1 inference is not bound by per-op host overhead.
 *
 * enc_tab_host: host array of device pointers, 3 per conv layer {w, scale, shift}, in order
 *   h1, h2, 32 residual convs (res_block_enc_0/enc_0_1/conv1 ... res_block_enc_final/conv2),
 *   to_bn; then centers.  Count = 3*35 + 1 = 106.  The 32 residual filters are PACKED with
 *   ic_pack_conv3x3_c128_f32, h2 and to_bn with ic_pack_conv2d_mfma_f32; h1 is TF layout.
 * dec_tab_host: from_bn, 32 residual convs, h12, h13 -> 3*35 = 105 pointers (from_bn and h13 filters in TF
 *   conv2d_transpose layout, residual filters packed, h12 packed with ic_pack_conv2d_mfma_f32).
 * B = arch_param_B (5).  C = num_chan_bn.  x: (N,3,H,W) float 0..255, H and W multiples of 8.
 * Outputs of encode (each nullable except symbols/qhard): heatmap,z,qsoft,qhard,qbar (N,C,H/8,W/8),
 * symbols int64.  decode: q (N,C,H/8,W/8) -> x_out (N,3,H,W) clipped to [0,255].
 * flags: IC_CONV3_* for the 64 3x3 launches of the call (e.g. IC_CONV3_LEAVE_IDLE_CUS while an independent branch runs
 * on the CUs a partly filled round leaves free).
 */
size_t ic_ae_workspace_bytes(int N, int H, int W, int C);
/* With IC_CONV3_STACK_KERNEL in `flags`, a residual stack whose NB-segment jobs fit the chip in one round (a Kodak map, a
 * 256 x 256 image) runs its 6B+2 layers as ONE persistent launch whose work-groups hand activations to their neighbours
 * through flags in the workspace (conv3x3_wino_stack.hip; outputs are bit-identical to the per-layer launches).
 * Byte offset of that sync area inside the workspace: its first 32-bit word is 0 after a call unless a hand-off timed out
 * (1 + layer index; the call's outputs are invalid then).  reference: the hot loop is one image per run, val.py:157-158. */
size_t ic_ae_sync_pos_bytes(int N, int H, int W, int C);
int ic_ae_encode_f32(const float* x, const void* const* enc_tab_host, int B, int C, int L, int heatmap_on,
                     int normalize_on, float* heatmap, float* z, float* qsoft, float* qhard, float* qbar,
                     int64_t* symbols, int N, int H, int W,
                     void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream);
int ic_ae_decode_f32(const float* q, const void* const* dec_tab_host, int B, int C, int normalize_on,
                     float* x_out, int N, int H, int W,
                     void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream);
/* the residual stack alone (autoencoder.py:224-234 / :252-262): x (N,128,H,W) -> y, the 6B+2 3x3 layers of one table
 * (3 pointers per layer, as above); what bench.py times the dominant kernel with, in its real launch sequence.
 * workspace: ic_ae_res_stack_workspace_bytes(N, H, W). */
size_t ic_ae_res_stack_workspace_bytes(int N, int H, int W);
size_t ic_ae_res_stack_sync_pos_bytes(int N, int H, int W);      /* as ic_ae_sync_pos_bytes, for this entry point */
int ic_ae_res_stack_f32(const float* x, const void* const* tab_host, int B, float* y, int N, int H, int W,
                        void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream);

/* ic_bn_stats_f32 plus everything the training loop folds from it, in the same two launches:
 *   mean, invstd = 1/sqrt(var + eps), scale = gamma * invstd, shift = beta - mean * scale   (all [C], outputs)
 *   moving_mean / moving_var (optional, updated in place): m = m * decay + stat * (1 - decay), the variance fed to the
 *   moving average being the unbiased one (slim.batch_norm fused=True, autoencoder.py:106-125).
 * workspace: ic_bn_workspace_bytes(C). */
int ic_bn_train_stats_f32(const float* x, const float* gamma, const float* beta, float* moving_mean,
                          float* moving_var, float decay, float eps, float* mean, float* invstd, float* scale,
                          float* shift, int N, int C, int HW, void* workspace, ic_stream_t stream);
/* One layer's training-mode BatchNorm forward (slim.batch_norm is_training=True, autoencoder.py:106-125) in two launches:
 * ic_bn_train_stats_f32 followed by ic_bn_apply_f32 (y = act(x * scale + shift) + res1 + res2), bit for bit -- the element-wise
 * kernel folds the statistics from the partial sums itself instead of waiting for a fold kernel. */
int ic_bn_train_forward_f32(const float* x, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                            float decay, float eps, float* mean, float* invstd, float* scale, float* shift,
                            const float* res1, const float* res2, float* y, int N, int C, int HW, int relu,
                            void* workspace, ic_stream_t stream);
/* The same layer in ONE launch when the convolution that produced x left its per-segment channel sums behind (conv_stats
 * [C][parts][2], ic_wino4_3x3_c128_raw_stats_f32): summed in double in index order by every work-group of the channel, then folded
 * and applied exactly as above.  No workspace. */
int ic_bn_train_forward_cstats_f32(const float* x, const float* conv_stats, int parts, const float* gamma, const float* beta,
                                   float* moving_mean, float* moving_var, float decay, float eps, float* mean, float* invstd,
                                   float* scale, float* shift, const float* res1, const float* res2, float* y, int N, int C,
                                   int HW, int relu, ic_stream_t stream);

/* =============================================================================================
 * Training (train.py:101-106, :303-349): training-mode BatchNorm, backward kernels.
 * Forward in training mode = the conv entry points above with scale = 1, shift = 0, relu = 0 ("raw" conv),
 * then ic_bn_stats_f32 + ic_bn_apply_f32.  Data gradients of the convs reuse the forward kernels (a conv's
 * data gradient is the transposed conv with the same TF filter array and vice versa; the 3x3 stride-1 case takes
 * ic_pack_conv3x3_c128_bwd_f32).  Filter gradients: ic_conv2d_wgrad_f32.
 * ============================================================================================= */
/* Workspace of the BatchNorm entry points (ic_bn_stats_f32, ic_bn_train_stats_f32, ic_bn_train_forward_f32, ic_bn_moments_f32,
 * ic_bn_backward_reduce_f32, ic_bn_backward_f32): each call is TWO launches that communicate through it -- the first writes per-block
 * partial sums [C][blocks][0..3], the second (the element-wise kernel included) re-reads them in every work-group.  A workspace
 * must therefore NOT be shared by calls that can overlap in time: one per stream (calls on one stream are ordered and may
 * share one, as imgcomp_cvpr_amd/training.py does with its single compute stream).  Sharing across streams is a silent race. */
size_t ic_bn_workspace_bytes(int C);
/* batch mean and BIASED variance per channel of x (N,C,HW) (autoencoder.py:114-125, is_training=True) */
int ic_bn_stats_f32(const float* x, float* mean, float* var, int N, int C, int HW, void* workspace, ic_stream_t stream);
/* y = act(x * scale[c] + shift[c]) + res1 + res2 */
int ic_bn_apply_f32(const float* x, const float* scale, const float* shift, const float* res1, const float* res2,
                    float* y, int N, int C, int HW, int relu, ic_stream_t stream);
/* BN(+ReLU) backward: g = dy * [x*scale+shift > 0 if relu]; dbeta = sum g; dgamma = sum g*xhat;
 * dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M), xhat = (x - mean)*invstd */
int ic_bn_backward_f32(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                       const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                       int N, int C, int HW, int relu, void* workspace, ic_stream_t stream);
/* Cross-replica ("sync") BatchNorm for data-parallel training: the reference normalises over its whole batch on one device
 * (autoencoder.py:115-125, ae_configs/cvpr/base: batch_size 30); with the batch split over ranks the caller sums the
 * per-channel float64 moments of every rank between the two halves of each pass (2 C doubles per layer and direction):
 *   forward : ic_bn_moments_f32 -> sums = {sum x [C], sum x^2 [C]}      -> all-reduce(sum) -> ic_bn_train_fold_moments_f32
 *             (count = elements per channel over ALL ranks) -> ic_bn_apply_f32
 *   backward: ic_bn_backward_reduce_f32 -> sums = {sum g [C], sum g xhat [C]}, dbeta / dgamma = this rank's sums (the
 *             gradient all-reduce averages them like every other parameter gradient) -> all-reduce(sum) of `sums`
 *             -> ic_bn_backward_apply_f32 (count as above)
 * With one rank and count = N * HW both paths are bit-identical to ic_bn_train_stats_f32 / ic_bn_backward_f32. */
int ic_bn_moments_f32(const float* x, double* sums, int N, int C, int HW, void* workspace, ic_stream_t stream);
int ic_bn_train_fold_moments_f32(const double* sums, long long count, const float* gamma, const float* beta,
                                 float* moving_mean, float* moving_var, float decay, float eps, float* mean,
                                 float* invstd, float* scale, float* shift, int C, ic_stream_t stream);
int ic_bn_backward_reduce_f32(const float* dy, const float* x, const float* scale, const float* shift,
                              const float* mean, const float* invstd, double* sums, float* dgamma, float* dbeta,
                              int N, int C, int HW, int relu, void* workspace, ic_stream_t stream);
int ic_bn_backward_apply_f32(const float* dy, const float* x, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* gamma, const double* sums,
                             long long count, float* dx, int N, int C, int HW, int relu, ic_stream_t stream);
/* filter gradient, generic form dW[t][a][b] = sum U[n][a][s*q + t + o0] * V[n][b][q] (conv_wgrad.hip):
 *   conv:            U = x (A = Cin, H x W), V = dy (B = Cout)            -> dw [KH][KW][Cin][Cout]
 *   transposed conv: U = dy (A = Cout, 2H x 2W), V = x (B = Cin), stride 2 -> dw [KH][KW][Cout][Cin]
 * w (nullable) and wd add the L2 term wd * w (slim.l2_regularizer, autoencoder.py:101-102). */
size_t ic_conv2d_wgrad_workspace_bytes(int N, int A, int B, int VH, int VW, int KH, int KW);
int ic_conv2d_wgrad_f32(const float* U, const float* V, float* dw, int N, int A, int UH, int UW, int B,
                        int KH, int KW, int stride, const float* w, float wd,
                        void* workspace, size_t workspace_bytes, ic_stream_t stream);
/* Filter gradient of the 3x3 128 -> 128 layer in the Winograd domain (16/36 of the direct form's multiply-adds):
 *   dw[ky][kx][ci][co] = sum_{n,y,x} x[n][ci][y+ky-1][x+kx-1] * dy[n][co][y][x]  (+ wd * w),  TF filter layout, SAME padding.
 * Replaces tf.gradients of the residual blocks' slim.conv2d w.r.t. `weights` (autoencoder.py:274-287).  Even W only and tensors
 * below 2^31 bytes; IC_ERR_UNSUPPORTED otherwise (workspace query returns 0): use ic_conv2d_wgrad_f32 then. */
size_t ic_conv3x3_c128_wgrad_workspace_bytes(int N, int H, int W);
int ic_conv3x3_c128_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, const float* w, float wd,
                              void* workspace, size_t workspace_bytes, ic_stream_t stream);
int ic_pack_conv3x3_c128_bwd_f32(const float* w_tf, float* w_packed, ic_stream_t stream);
/* importance map + quantiser backward (autoencoder.py:127-134,171-200; quantizer.py:43-100): d_qbar, d_heatmap
 * (nullable) -> d_bottleneck (N,C+1,h,w), d_centers (L).  workspace: ic_heatmap_quantize_bwd_workspace_bytes(L). */
size_t ic_heatmap_quantize_bwd_workspace_bytes(int L);
int ic_heatmap_quantize_bwd_f32(const float* bottleneck, const float* centers, int L, float sigma,
                                const float* d_qbar, const float* d_heatmap, float* d_bottleneck, float* d_centers,
                                int N, int C, int h, int w, int heatmap_on, void* workspace, ic_stream_t stream);
/* context model backward pieces (probclass.py:63-106,185-261) */
int ic_pc_dlogits_f32(const float* logits, const int64_t* symbols, const float* d_bits, float* g,
                      int N, int vol, int L, ic_stream_t stream);
/* data gradient of one masked conv3d layer: dx = (sum_taps w^T g (+ res embedded at (2,2,2))) * [act > 0].
 * workspace = ic_pc_bwd_data_workspace_bytes(...) bytes runs it on the matrix cores (the layer's adjoint: mirrored taps,
 * transposed filter, zero-padded gradient); NULL / 0 or an uncovered shape (size query returns 0) runs the VALU kernel. */
size_t ic_pc_bwd_data_workspace_bytes(int N, int Cin, int Cout, int OD, int OH, int OW);
int ic_pc_bwd_data_f32(const float* g, const float* w, const float* res, const float* act, float* dx,
                       int N, int Cin, int Cout, int OD, int OH, int OW, int first_mask, int relu_mask,
                       void* workspace, size_t workspace_bytes, ic_stream_t stream);
size_t ic_pc_wgrad_workspace_bytes(int N, int A, int B, int VD, int VH, int VW);
int ic_pc_wgrad_f32(const float* U, const float* q, float pad_value, const float* V, float* dw,
                    int N, int A, int B, int VD, int VH, int VW, int first_mask,
                    void* workspace, size_t workspace_bytes, ic_stream_t stream);
size_t ic_channel_sum_workspace_bytes(int C);
int ic_channel_sum_f32(const float* x, float* out, int N, int C, int M, void* workspace, ic_stream_t stream);
/* tf.train.AdamOptimizer step on a flat bucket (train.py:339-349, training_helpers.py:38-48): var, m, v updated in place;
 * lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) from the host; epsilon outside the bias correction.  16-byte aligned pointers. */
int ic_adam_tf_f32(float* var, const float* grad, float* m, float* v, long long count, float lr_t, float beta1,
                   float beta2, float eps, ic_stream_t stream);
/* The context-model backward reads the three feature volumes ic_pc_bitcost_f32 left in its workspace
 * (layout: conv0 out (N,k,C+3,h+6,w+6) | res1/conv1 out (N,k,C+2,h+4,w+4) | res1 out (N,k,C+1,h+2,w+2) | packed
 * filters): keep that workspace untouched between the forward call and the ic_pc_* backward calls. */

/* Branch streams.  The context model and the decoder are independent consumers of the encoder output (val.py:85-89);
 * ic_stream_create_cu_range makes a stream whose kernels only run on CU mask bits [first_cu, first_cu + n_cus) -- bit i
 * is CU i/8 of XCD i%8 on the MI355X, so a multiple of 8 takes the same number of CUs from every XCD.  Giving the
 * context model the CUs the decoder's one-work-group-per-CU 3x3 launches leave idle (a Kodak map: 64 of 256) lets the
 * branches overlap fully.  The stream is blocking with respect to the legacy default stream: run the other branch on a
 * non-blocking stream.  IC_ERR_UNSUPPORTED if the range exceeds the device. */
int ic_stream_create_cu_range(int first_cu, int n_cus, ic_stream_t* stream);
int ic_stream_destroy(ic_stream_t stream);

/* device timing helper for bench.py: wall time between two points on `stream` measured with
 * hipEvents created on that stream's device (torch.cuda.Event only sees torch's own streams). */
int ic_event_create(void** ev);
int ic_event_destroy(void* ev);
int ic_event_record(void* ev, ic_stream_t stream);
int ic_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on `stop` */

/* ---------------------------------------------------------------------------------------------
 * Cross-replica sums over peer-mapped device memory (csrc/peer_exchange.hip): what cross-replica BatchNorm exchanges once per
 * layer and direction when one batch is split over the GPUs of a node (reference: single-device batch statistics,
 * autoencoder.py:106-125; train.py:150-153).  One process per GPU; the 64-byte handles travel between the processes by any
 * means (the Python host uses the process group).  Set-up calls allocate / map memory; ic_peer_allreduce_f64 is one launch.
 * --------------------------------------------------------------------------------------------- */
size_t ic_peer_region_bytes(void);
int ic_peer_max_values(void);      /* doubles per exchange */
int ic_peer_max_world(void);
int ic_peer_region_create(void** region, void* ipc_handle_64);        /* zeroed fine-grained region on the current device */
int ic_peer_region_open(const void* ipc_handle_64, void** mapped);    /* a peer's region, mapped into this process */
int ic_peer_region_close(void* mapped);
int ic_peer_region_destroy(void* region);
/* vals[n] (device, float64): in place -> the sum over `world` ranks in rank order (bit-identical on every rank).
 * regions_host[world]: every rank's region as mapped here (regions_host[rank] = own); seq = 1, 2, 3 ... the same on every rank;
 * status: device int, set to 1 if a peer's flag did not arrive within the spin bound (vals are left unchanged then). */
int ic_peer_allreduce_f64(double* vals, int n, void* const* regions_host, int rank, int world, unsigned seq, int* status,
                          ic_stream_t stream);
/* the same with an explicit bound on the polls of one peer flag (0 = the default, a few seconds): tests force the time-out
 * path with a small bound.  status must not be NULL (IC_ERR_ARG): the time-out path stores through it. */
int ic_peer_allreduce_f64_bounded(double* vals, int n, void* const* regions_host, int rank, int world, unsigned seq,
                                  unsigned spin_limit, int* status, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The same 3x3 128 -> 128 layer (autoencoder.py:274-287) in Winograd F(4x4,3x3) form (csrc/conv3x3_wino4.hip): 36 / 144 of the
 * direct form's multiplies.  W % 4 == 0 (ic_wino4_3x3_c128_supported); filters packed by ic_pack_wino4_3x3_c128_f32
 * (36 x 128 x 128 floats; backward = 1: the adjoint filter of the data gradient).  Same epilogue contract as
 * ic_wino3x3_c128_bn_act_f32: y = act(conv * scale + shift) + res1 + res2.
 * --------------------------------------------------------------------------------------------- */
/* the form ic_conv3x3_c128_auto_f32 (and with it the whole-network entry points) runs for a shape and flags:
 * 0 direct, 1 Winograd F(2x2,3x3), 2 Winograd F(4x4,3x3) */
int ic_conv3x3_c128_pick_form(int N, int H, int W, int flags);
size_t ic_wino4_3x3_c128_packed_floats(void);
int ic_pack_wino4_3x3_c128_f32(const float* w_tf, float* w_packed, int backward, ic_stream_t stream);
/* all 3x3 filters of a network in one launch: w_tf_table_dev (device array of device pointers) -> fragments l at w_packed + l * packed_floats */
int ic_pack_wino4_3x3_c128_batch_f32(const float* const* w_tf_table_dev, float* w_packed, int layers, int backward, ic_stream_t stream);
int ic_wino4_3x3_c128_supported(int N, int H, int W);
long long ic_wino4_3x3_c128_workgroups(int N, int H, int W);     /* in units of the 4-wave form: 2 per segment of 16 tiles */
/* waves per work-group the launch takes for this shape and flags: 4 (two 64-channel work-groups per segment, two per CU) or 8 (one
 * work-group per segment and CU: IC_CONV3_WINO4_WG8, or by itself from 2048 four-wave work-groups on); 0: shape not supported.
 * Both forms give the same bits (profiles/r06_w4_wg8.md has the measurements). */
int ic_wino4_3x3_c128_waves(int N, int H, int W, int flags);
int ic_wino4_3x3_c128_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                 const float* res1, const float* res2, float* y, int N, int H, int W, int relu,
                                 int flags, ic_stream_t stream);
/* The training step's forward convolution (train.py:101-106 runs the graph with is_training=True): y = the RAW convolution (no
 * BatchNorm fold, activation or residual), and per output channel c and segment s the sum and the sum of squares of the values
 * stored -- stats[(c * parts + s) * 2 + {0, 1}], parts = ic_wino4_3x3_c128_stats_parts(N, H, W), fp32 over <= 256 values each, in
 * a fixed order (bit-reproducible).  ic_bn_train_forward_cstats_f32 makes the layer's batch statistics of them: training-mode
 * BatchNorm (autoencoder.py:106-125) without a pass over y for the statistics. */
long long ic_wino4_3x3_c128_stats_parts(int N, int H, int W);
int ic_wino4_3x3_c128_raw_stats_f32(const float* x, const float* w_packed, float* y, float* stats, int N, int H, int W,
                                    int flags, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The two large 5x5 / stride-2 layers on the same kernel (csrc/conv3x3_wino4.hip): a 5x5 / stride-2 SAME convolution is ONE 3x3 /
 * stride-1 convolution over the four phases of its input stacked as channels, the transposed one is ONE 3x3 convolution to four
 * phase planes per output channel (tools/phase_conv_check.py) -- in F(4x4,3x3) form 3.62 instead of 10.07 GFLOP per Kodak image.
 *   h2  (slim.conv2d 64 -> 128, [5,5], stride 2, BN, ReLU; autoencoder.py:223):
 *        x_phases [N][4][64][H][W]: plane (2 py + px, c) holds X[c][2 i + py][2 j + px] of the [N][64][2H][2W] input
 *        (ic_space_to_depth2_f32 makes it; ic_ae_encode_f32 has h1 write it directly) -> y [N][128][H][W]
 *   h12 (slim.conv2d_transpose 128 -> 64, [5,5], stride 2, BN, ReLU; autoencoder.py:264):  x [N][128][H][W] -> y [N][64][2H][2W]
 * Filters: ic_pack_wino4_conv5s2_f32 from the TF arrays ([5][5][64][128] for both: conv [kh][kw][in][out], transposed
 * [kh][kw][out][in]), 36 x 256 x 128 floats.  W % 4 == 0 and 256 H W floats below 2 GiB (ic_wino4_conv5s2_supported).
 * --------------------------------------------------------------------------------------------- */
size_t ic_wino4_conv5s2_packed_floats(void);
int ic_pack_wino4_conv5s2_f32(const float* w_tf, float* w_packed, int transposed, ic_stream_t stream);
int ic_wino4_conv5s2_supported(int N, int H, int W);
long long ic_wino4_conv5s2_workgroups(int N, int H, int W, int transposed);
int ic_wino4_conv5s2_c64_c128_bn_act_f32(const float* x_phases, const float* w_packed, const float* scale, const float* shift,
                                         float* y, int N, int H, int W, int relu, int flags, ic_stream_t stream);
int ic_wino4_deconv5s2_c128_c64_bn_act_f32(const float* x, const float* w_packed, const float* scale, const float* shift,
                                           float* y, int N, int H, int W, int relu, int flags, ic_stream_t stream);
/* both filter forms of h2 (transposed = 0) / h12 (transposed = 1) in one blob: ic_pack_conv2d_mfma_f32's fragments, then
 * ic_pack_wino4_conv5s2_f32's at float offset ic_conv2d_mfma_packed_floats(5, 5, cin, cout, 2, transposed) */
size_t ic_conv5s2_both_packed_floats(int transposed);
int ic_pack_conv5s2_both_f32(const float* w_tf, float* w_packed, int transposed, ic_stream_t stream);
/* x [N][C][H2][W2] (H2, W2 even) -> y [N][4][C][H2/2][W2/2], plane 2 py + px = the pixels (2 i + py, 2 j + px) */
int ic_space_to_depth2_f32(const float* x, float* y, int N, int C, int H2, int W2, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * val.py's image metrics (val.py:92-96, ms_ssim_np.py:49-110) of two uint8 NCHW batches, float64 on the device (csrc/val_metrics.hip):
 * out6[s] = mean contrast-structure term of scale s (s = 0 .. 3), out6[4] = mean SSIM of scale 4, out6[5] = mean squared error.
 * MS-SSIM = prod_s out6[s] ^ w_s with ms_ssim_np.py's weights, PSNR = 10 log10(255^2 / out6[5]) (imgcomp_cvpr_amd/metrics.py).
 * Every scale needs at least one 'valid' blur output (any image of 16 x 16 pixels or more).
 * --------------------------------------------------------------------------------------------- */
size_t ic_val_metrics_workspace_bytes(int N, int C, int H, int W);
int ic_val_metrics_u8_f64(const unsigned char* x, const unsigned char* y, int N, int C, int H, int W, double* out6,
                          void* workspace, size_t workspace_bytes, ic_stream_t stream);
/* ... for every image of the batch separately: image n's six values at out6n + 6 n (the single-image sequence N times on `stream`,
 * bit-identical to N calls with N = 1); workspace: ic_val_metrics_workspace_bytes(1, C, H, W) */
int ic_val_metrics_per_image_u8_f64(const unsigned char* x, const unsigned char* y, int N, int C, int H, int W, double* out6n,
                                    void* workspace, size_t workspace_bytes, ic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * MS-SSIM training distortion and its gradient (csrc/msssim.hip).  Reference: code/ms_ssim.py:3-186 (5 scales, separable
 * Gaussian 'VALID' blur with REFLECT pads on small scales, 2x2 box between scales), train.py:352-394
 * (d_loss_scaled = K_ms_ssim * (1 - MS-SSIM(x, x_out)); TensorFlow derives d / d x_out, here it is written out).
 * The blur matrices of a shape are computed once on the HOST (ic_msssim_plan_fill -> a plain buffer the caller uploads) and
 * passed as a device pointer with every call.  IC_ERR_UNSUPPORTED / 0 bytes: an image too small for five scales.
 * --------------------------------------------------------------------------------------------- */
size_t ic_msssim_plan_bytes(int H, int W);
int ic_msssim_plan_fill(int H, int W, void* host_buf, size_t bytes);          /* CPU arithmetic only */
size_t ic_msssim_workspace_bytes(int N, int C, int H, int W);
/* x, x_out: (N,C,H,W) float32 in 0..255.  grad_out (N,C,H,W) = d (K (1 - MS-SSIM)) / d x_out (NULL: value only).
 * scalars_out: 16 device floats: [0] MS-SSIM, [1] K (1 - MS-SSIM), [2..5] cs of scales 0..3, [6] ssim of scale 4,
 * [8..12] dL/dS_l / positions_l.  Means are reduced in float64 in a fixed order (bit-reproducible). */
int ic_msssim_loss_grad_f32(const float* x, const float* x_out, int N, int C, int H, int W, float K, const void* plan_dev,
                            float* grad_out, float* scalars_out, void* workspace, size_t workspace_bytes, ic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMGCOMP_HIP_H */
