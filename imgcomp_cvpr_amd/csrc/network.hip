// Whole-network entry points for the CVPR autoencoder + library plumbing (version, errors, events).
//   reference: code/autoencoder.py:218-244 (_CVPR._encode), :246-268 (_CVPR._decode)
// One host call enqueues all 36 stages of encode (or decode) on the caller's stream: batch-1 inference
// (val.py runs one image per step, code/val.py:157-158) is otherwise bound by per-op host overhead.
#include "internal.h"
#include "wino_common.h"
#include <string.h>

// ---- residual stack shared by encoder and decoder (autoencoder.py:224-234 / :252-262) ----
// tab: 3 pointers {packed filter (both forms, ic_pack_conv3x3_c128_both_f32), scale, shift} per conv, 6B+2 convs.  bufs[0] holds the stack input
// (kept for the global skip), bufs[4] is the temporary.  Returns the buffer index holding the output.
// flags: IC_CONV3_* for every launch; IC_CONV3_LEAVE_IDLE_LAYERS(n) limits IC_CONV3_LEAVE_IDLE_CUS to the first n launches
// (a caller's side branch that needs its CUs for less than the whole stack).
static inline int layer_flags(int flags, int li) {
    const int n = (flags >> 12) & 0x7f;
    int f = flags & (0xfff | IC_CONV3_IN_FLIGHT(0xf) | IC_CONV3_NO_WINO4 | IC_CONV3_WINO4_BITS);
    if (n > 0 && li >= n) f &= ~IC_CONV3_LEAVE_IDLE_CUS;
    return f;
}

// One layer of the stack as the walk below visits it: the per-layer path launches it, the persistent path records it.
struct StackWalk {
    bool record; WnStackArgs* sa; int flags; int N, H, W; hipStream_t st; int li; int nlayers;
    int conv(const float* x, const float* const* l, const float* r1, const float* r2, float* y, int relu) {
        if (record) {
            WnStackLayer& L = sa->layers[li];
            L.x = x; L.wp = l[0] + ic_conv3x3_c128_packed_floats(); L.scale = l[1]; L.shift = l[2];
            L.res1 = r1; L.res2 = r2; L.y = y; L.relu = relu; L.pad_ = 0;
            ++li;
            return IC_OK;
        }
        // (the table holds the stack's layers one after the other: l + 3 is the next layer's entry)
        const int rc = icx_conv3x3_c128_auto_next(x, l[0], l[1], l[2], r1, r2, y, N, H, W, relu, layer_flags(flags, li),
                                                  li + 1 < nlayers ? l[3] : nullptr, st);
        ++li;
        return rc;
    }
};

static int res_stack_walk(StackWalk& w, const void* const* tab, int B, float* const bufs[5], int* out_idx) {
    int cur = 0, rc;
    float* T = bufs[4];
    const float* const* t = (const float* const*)tab;
    for (int b = 0; b < B; ++b) {
        const int G = cur;
        for (int i = 0; i < 3; ++i) {
            int O = 1;
            while (O == G || O == cur) ++O;              // one of {1,2,3} is always free
            const float* const* l1 = t + 3 * w.li;
            if ((rc = w.conv(bufs[cur], l1, nullptr, nullptr, T, 1))) return rc;
            if ((rc = w.conv(T, l1 + 3, bufs[cur], i == 2 ? bufs[G] : nullptr, bufs[O], 0))) return rc;
            cur = O;
        }
    }
    // final block: both convs linear, + block input + stack input
    {
        int O = 1;
        while (O == cur) ++O;
        const float* const* l1 = t + 3 * w.li;
        if ((rc = w.conv(bufs[cur], l1, nullptr, nullptr, T, 0))) return rc;
        if ((rc = w.conv(T, l1 + 3, bufs[cur], bufs[0], bufs[O], 0))) return rc;
        cur = O;
    }
    *out_idx = cur;
    return IC_OK;
}

// Segments per job if this shape's whole stack can run as ONE persistent launch (conv3x3_wino_stack.hip), else 0: the per-layer
// plan must be NB-segment jobs only, in one resident round, and the caller must not have asked for a particular form, for idle
// CUs (a concurrent branch on a CU-range stream could keep work-groups of a persistent launch from becoming resident together)
// -- and must have asked for it (IC_CONV3_STACK_KERNEL): measured, the hand-off costs what the kernel boundary costs.
static int stack_kernel_nb(int N, int H, int W, int nlayers, int flags) {
#ifndef IC_TUNING
    return 0;                                            // the persistent stack kernel is compiled into tuning builds only
#else
    if (!(flags & IC_CONV3_STACK_KERNEL) || (flags & (IC_CONV3_LEAVE_IDLE_CUS | IC_CONV3_PACKED_TRANSFORM))) return 0;
    // several calls in flight: a persistent launch needs ALL its work-groups resident at once, which concurrent launches of other
    // images cannot guarantee -- hand-offs would run into their spin bound
    if (((flags >> 19) & 0xf) >= 2) return 0;
    const int form = flags & IC_CONV3_FORM_MASK;
    if (form != IC_CONV3_AUTO && form != IC_CONV3_WINO && !(form >= IC_CONV3_WINO_SEG1 && form <= IC_CONV3_WINO_SEG3)) return 0;
    if (ic_conv3x3_c128_pick_algo(N, H, W, flags) != 1) return 0;
    long long pl[5];
    if (ic_wino3x3_c128_plan(N, H, W, flags & IC_CONV3_FORM_MASK, pl) != IC_OK) return 0;
    if (pl[0] || pl[3] || pl[4] || !pl[1]) return 0;
    return icx_wino_stack_fits(N, H, W, (int)pl[2], nlayers) ? (int)pl[2] : 0;
#endif
}

// sync: WN_STACK_SYNC_BYTES of the caller's workspace for the persistent form (nullptr: per-layer launches only)
static int res_stack(const void* const* tab, int B, float* const bufs[5], int N, int H, int W, int flags,
                     hipStream_t st, int* out_idx, unsigned* sync) {
    const int nlayers = 6 * B + 2;
    const int nb = sync ? stack_kernel_nb(N, H, W, nlayers, flags) : 0;
#ifdef IC_TUNING
    if (nb) {
        WnStackArgs sa{};
        StackWalk w{true, &sa, flags, N, H, W, st, 0, nlayers};
        int rc = res_stack_walk(w, tab, B, bufs, out_idx);
        if (rc) return rc;
        sa.flags = sync; sa.N = N; sa.H = H; sa.W = W; sa.grows = ic_cdiv(H, 4); sa.gcols = ic_cdiv(W, 32);
        sa.nlayers = nlayers; sa.xcd_runs = (flags & IC_CONV3_NO_XCD_RUNS) ? 0 : 1;
        return icx_wino_stack_launch(sa, nb, st);
    }
#endif
    (void)nb;
    StackWalk w{false, nullptr, flags, N, H, W, st, 0, nlayers};
    return res_stack_walk(w, tab, B, bufs, out_idx);
}

static size_t ae_ws_floats(int N, int H, int W, int C) {
    const size_t hw = (size_t)H * W;
    // 5 x (N,128,H/4,W/4) + (N,64,H/2,W/2) + bottleneck (N,C+1,H/8,W/8)
    return (size_t)N * (5 * 8 * hw + 16 * hw + (size_t)(C + 1) * hw / 64);
}
// the persistent residual-stack kernel's sync area sits behind the activations, on a 256-byte boundary
static size_t ae_sync_offset(int N, int H, int W, int C) { return (ae_ws_floats(N, H, W, C) * sizeof(float) + 255) & ~(size_t)255; }

extern "C" size_t ic_ae_workspace_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return ae_sync_offset(N, H, W, C) + WN_STACK_SYNC_BYTES;
}
// byte offset, inside a workspace of ic_ae_workspace_bytes, of the sync area of the persistent residual-stack kernel; its first
// 32-bit word is 0 after a call unless a hand-off between work-groups timed out (then: 1 + the layer; the outputs are invalid)
extern "C" size_t ic_ae_sync_pos_bytes(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return ae_sync_offset(N, H, W, C);
}

static void carve(void* ws, int N, int H, int W, float* bufs[5], float** half, float** bott) {
    const size_t hw = (size_t)H * W;
    float* p = (float*)ws;
    for (int i = 0; i < 5; ++i) { bufs[i] = p; p += (size_t)N * 8 * hw; }
    *half = p; p += (size_t)N * 16 * hw;
    *bott = p;
}

// The residual stack alone (autoencoder.py:224-234 / :252-262): x (N,128,H,W) -> y (N,128,H,W), 6B+2 convs.  tab as in
// ic_ae_encode_f32 for these layers only (3 pointers per conv).  Used by bench.py to time the dominant kernel in-step and by
// tests; workspace: 5 x N x 128 x H x W floats (bufs[0] receives a copy of x).
extern "C" size_t ic_ae_res_stack_workspace_bytes(int N, int H, int W) {
    return N > 0 && H > 0 && W > 0 ? (size_t)5 * N * 128 * H * W * sizeof(float) + WN_STACK_SYNC_BYTES : 0;
}
extern "C" size_t ic_ae_res_stack_sync_pos_bytes(int N, int H, int W) {
    return N > 0 && H > 0 && W > 0 ? (size_t)5 * N * 128 * H * W * sizeof(float) : 0;
}
extern "C" int ic_ae_res_stack_f32(const float* x, const void* const* tab, int B, float* y, int N, int H, int W,
                                   void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && tab && y && workspace && N > 0 && H > 0 && W > 0 && B >= 0);
    if (workspace_bytes < ic_ae_res_stack_workspace_bytes(N, H, W)) return IC_ERR_WORKSPACE;
    for (int i = 0; i < 3 * (6 * B + 2); ++i) IC_CHECK_ARG(tab[i] != nullptr);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)N * 128 * H * W;
    float* bufs[5];
    for (int i = 0; i < 5; ++i) bufs[i] = (float*)workspace + i * n;
    hipError_t e = hipMemcpyAsync(bufs[0], x, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    int o, rc;
    if ((rc = res_stack(tab, B, bufs, N, H, W, flags, st, &o, (unsigned*)((char*)workspace + ic_ae_res_stack_sync_pos_bytes(N, H, W))))) return rc;
    e = hipMemcpyAsync(y, bufs[o], n * sizeof(float), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? IC_OK : (int)e;
}

// ---- h2 / h12: direct MFMA kernels or the F(4x4)-over-phases form (conv3x3_wino4.hip) ----
// The caller says with IC_CONV5_BOTH_PACKED that the two filter blobs carry both fragment sets; the F(4x4) form runs where the residual
// stack of the same call runs F(4x4) (same map, at least as many work-groups) or the map is at least Kodak-sized, or wherever the
// shape allows with IC_CONV5_WINO4.
static bool edge_layers_wino4(int N, int H4, int W4, int flags) {
    if (!(flags & IC_CONV5_BOTH_PACKED) || (flags & IC_CONV5_NO_WINO4) || !ic_wino4_conv5s2_supported(N, H4, W4)) return false;
    if (flags & IC_CONV5_WINO4) return true;
    // (>= 160 work-groups of h2's launch, i.e. from one Kodak map on, also one image at a time: 85 + 93 -> 53 + 51 us alone, the
    // one-at-a-time step 159.8 -> 163.0 Mpix/s; smaller maps keep the direct kernels: 28 / 39 against 49 / 44 us at 64 x 64)
    return ic_conv3x3_c128_pick_form(N, H4, W4, flags) == 2 || ic_wino4_conv5s2_workgroups(N, H4, W4, 0) >= 160;
}
extern "C" size_t ic_conv5s2_both_packed_floats(int transposed) {
    return ic_conv2d_mfma_packed_floats(5, 5, transposed ? 128 : 64, transposed ? 64 : 128, 2, transposed ? 1 : 0) + ic_wino4_conv5s2_packed_floats();
}
extern "C" int ic_pack_conv5s2_both_f32(const float* w_tf, float* w_packed, int transposed, ic_stream_t stream) {
    IC_CHECK_ARG(w_tf && w_packed);
    const int tr = transposed ? 1 : 0, cin = tr ? 128 : 64, cout = tr ? 64 : 128;
    const int rc = ic_pack_conv2d_mfma_f32(w_tf, w_packed, 5, 5, cin, cout, 2, tr, stream);
    if (rc != IC_OK) return rc;
    return ic_pack_wino4_conv5s2_f32(w_tf, w_packed + ic_conv2d_mfma_packed_floats(5, 5, cin, cout, 2, tr), tr, stream);
}

extern "C" int ic_ae_encode_f32(const float* x, const void* const* tab, int B, int C, int L, int heatmap_on,
                                int normalize_on, float* heatmap, float* z, float* qsoft, float* qhard, float* qbar,
                                int64_t* symbols, int N, int H, int W,
                                void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(x && tab && workspace && N > 0 && H > 0 && W > 0 && B >= 0 && C > 0 && L > 0);
    if (H % 8 || W % 8) return IC_ERR_UNSUPPORTED;          // callers pad to the subsampling factor (val.py:157)
    if (workspace_bytes < ic_ae_workspace_bytes(N, H, W, C)) return IC_ERR_WORKSPACE;
    const int nconv = 6 * B + 2;
    for (int i = 0; i < 3 * (nconv + 3) + 1; ++i) IC_CHECK_ARG(tab[i] != nullptr);
    hipStream_t st = (hipStream_t)stream;
    float* bufs[5]; float* half; float* bott;
    carve(workspace, N, H, W, bufs, &half, &bott);
    int rc;
    const float* const* t = (const float* const*)tab;
    ConvArgs a{};
    // h1: 3 -> 64, 5x5 / 2, BN, ReLU, input normalisation folded into the load
    a.x = x; a.w = t[0]; a.scale = t[1]; a.shift = t[2]; a.y = half;
    a.N = N; a.Cin = 3; a.H = H; a.W = W; a.Cout = 64; a.KH = 5; a.KW = 5; a.stride = 2; a.relu = 1;
    a.builtin_norm = normalize_on ? 1 : 0;
    // h2: 64 -> 128, 5x5 / 2, BN, ReLU: F(4x4) over the four phases of h1's output (h1 then writes phase planes), or the direct form
    rc = IC_ERR_UNSUPPORTED;
    if (edge_layers_wino4(N, H / 4, W / 4, flags)) {
        a.out_phases = 1;
        rc = icx_conv2d(a, false, st);
        if (rc == IC_OK)
            rc = ic_wino4_conv5s2_c64_c128_bn_act_f32(half, t[3] + ic_conv2d_mfma_packed_floats(5, 5, 64, 128, 2, 0), t[4], t[5], bufs[0],
                                                      N, H / 4, W / 4, 1, flags, stream);
        else if (rc != IC_ERR_UNSUPPORTED) return rc;
        a.out_phases = 0;
    }
    if (rc == IC_ERR_UNSUPPORTED) {
        if ((rc = icx_conv2d(a, false, st))) return rc;
        rc = ic_conv2d_mfma_bn_act_f32(half, t[3], t[4], t[5], bufs[0], N, 64, H / 2, W / 2, 128, 5, 5, 2, 0, 1, st);
    }
    if (rc) return rc;
    int o;
    if ((rc = res_stack(tab + 6, B, bufs, N, H / 4, W / 4, flags, st, &o, (unsigned*)((char*)workspace + ae_sync_offset(N, H, W, C))))) return rc;
    // to_bn: 128 -> C(+1), 5x5 / 2, BN, linear
    const float* const* tb = t + 6 + 3 * nconv;
    const int Cb = C + (heatmap_on ? 1 : 0);
    if ((rc = ic_conv2d_mfma_bn_act_f32(bufs[o], tb[0], tb[1], tb[2], bott, N, 128, H / 4, W / 4, Cb, 5, 5, 2, 0, 0, st)))
        return rc;
    const float* centers = tb[3];
    if (heatmap_on)
        return ic_heatmap_quantize_f32(bott, centers, L, 1.0f, heatmap, z, qsoft, qhard, qbar, symbols,
                                       N, C, H / 8, W / 8, stream);
    // no importance map: z is the bottleneck itself
    const long long cnt = (long long)N * C * (H / 8) * (W / 8);
    if (z) {
        hipError_t e = hipMemcpyAsync(z, bott, cnt * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    if (qbar) return IC_ERR_UNSUPPORTED;   // qbar == qhard in value; request qhard instead
    if (!qsoft && !qhard && !symbols) return IC_OK;      // a network built with quantize=False (autoencoder.py:127-129): z only
    return ic_quantize_f32(bott, centers, L, 1.0f, qsoft, qhard, symbols, cnt, stream);
}

extern "C" int ic_ae_decode_f32(const float* q, const void* const* tab, int B, int C, int normalize_on,
                                float* x_out, int N, int H, int W,
                                void* workspace, size_t workspace_bytes, int flags, ic_stream_t stream) {
    IC_CHECK_ARG(q && tab && x_out && workspace && N > 0 && H > 0 && W > 0 && B >= 0 && C > 0);
    if (H % 8 || W % 8) return IC_ERR_UNSUPPORTED;
    if (workspace_bytes < ic_ae_workspace_bytes(N, H, W, C)) return IC_ERR_WORKSPACE;
    const int nconv = 6 * B + 2;
    for (int i = 0; i < 3 * (nconv + 3); ++i) IC_CHECK_ARG(tab[i] != nullptr);
    hipStream_t st = (hipStream_t)stream;
    float* bufs[5]; float* half; float* bott;
    carve(workspace, N, H, W, bufs, &half, &bott);
    int rc;
    const float* const* t = (const float* const*)tab;
    ConvArgs a{};
    // from_bn: C -> 128, 3x3 transposed / 2, BN, ReLU
    a.x = q; a.w = t[0]; a.scale = t[1]; a.shift = t[2]; a.y = bufs[0];
    a.N = N; a.Cin = C; a.H = H / 8; a.W = W / 8; a.Cout = 128; a.KH = 3; a.KW = 3; a.relu = 1;
    if ((rc = icx_conv2d(a, true, st))) return rc;
    int o;
    if ((rc = res_stack(tab + 3, B, bufs, N, H / 4, W / 4, flags, st, &o, (unsigned*)((char*)workspace + ae_sync_offset(N, H, W, C))))) return rc;
    const float* const* th = t + 3 + 3 * nconv;
    // h12: 128 -> 64, 5x5 transposed / 2, BN, ReLU
    rc = IC_ERR_UNSUPPORTED;
    if (edge_layers_wino4(N, H / 4, W / 4, flags))
        rc = ic_wino4_deconv5s2_c128_c64_bn_act_f32(bufs[o], th[0] + ic_conv2d_mfma_packed_floats(5, 5, 128, 64, 2, 1), th[1], th[2], half,
                                                    N, H / 4, W / 4, 1, flags, stream);
    if (rc == IC_ERR_UNSUPPORTED)
        rc = ic_conv2d_mfma_bn_act_f32(bufs[o], th[0], th[1], th[2], half, N, 128, H / 4, W / 4, 64, 5, 5, 2, 1, 1, st);
    if (rc) return rc;
    // h13: 64 -> 3, 5x5 transposed / 2, BN, linear, de-normalise, clip to [0, 255]
    a = ConvArgs{};
    a.x = half; a.w = th[3]; a.scale = th[4]; a.shift = th[5]; a.y = x_out;
    a.N = N; a.Cin = 64; a.H = H / 2; a.W = W / 2; a.Cout = 3; a.KH = 5; a.KW = 5; a.relu = 0;
    a.builtin_norm = normalize_on ? 2 : 4;   // 4: clip only (normalization = OFF still clips, autoencoder.py:267)
    return icx_conv2d(a, true, st);
}

// ---- plumbing ----
extern "C" int ic_abi_version(void) { return IC_ABI_VERSION; }

extern "C" const char* ic_strerror(int code) {
    switch (code) {
        case IC_OK: return "ok";
        case IC_ERR_ARG: return "invalid argument (null pointer or non-positive extent)";
        case IC_ERR_UNSUPPORTED: return "unsupported shape or option";
        case IC_ERR_WORKSPACE: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// CRC-32C (Castagnoli) of a HOST buffer: the checksum of TF-1 checkpoint tensors and table blocks
// (tensorflow/core/lib/hash/crc32c.h; saver.py:46-100 writes them, tf_checkpoint.py reads/writes them).
// Slicing-by-8 on the host; not a device function.
struct CrcTab { uint32_t t[8][256]; };
static constexpr CrcTab make_crc_tab() {
    CrcTab r{};
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        r.t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) r.t[t][i] = (r.t[t - 1][i] >> 8) ^ r.t[0][r.t[t - 1][i] & 0xff];
    return r;
}
// built at compile time: a constant of the library, no lazily initialised process-wide state
static constexpr CrcTab g_crc = make_crc_tab();
#define g_crc_tab g_crc.t

extern "C" uint32_t ic_crc32c(const void* data, size_t n, uint32_t crc) {
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = crc ^ 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = g_crc_tab[7][lo & 0xff] ^ g_crc_tab[6][(lo >> 8) & 0xff] ^ g_crc_tab[5][(lo >> 16) & 0xff] ^ g_crc_tab[4][lo >> 24] ^
            g_crc_tab[3][hi & 0xff] ^ g_crc_tab[2][(hi >> 8) & 0xff] ^ g_crc_tab[1][(hi >> 16) & 0xff] ^ g_crc_tab[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// ---- branch streams --------------------------------------------------------------------------------------------------
// The context model and the decoder both hang off the encoder output (val.py:85-89).  The decoder's 3x3 launches keep
// one 512-register work-group per CU and, for a Kodak-sized map, occupy 192 of the 256 CUs; a context-model work-group
// that lands on one of those CUs takes registers the next 3x3 work-group needs, and that one then waits for the whole
// SIMD.  A stream restricted to the CUs the decoder leaves idle removes the interference: the two branches overlap
// completely (round-1 A/B: 3.15 -> 2.98 ms per Kodak image).  Mask bit i is CU (i / 8) of XCD (i % 8)
// (tools/cumask_probe.hip), so a run of 8 m consecutive bits takes m CUs from every XCD.
// The stream is created BLOCKING (the only flavour the runtime offers with a mask): it orders itself against the legacy
// default stream, so the other branch must run on a non-blocking stream for the two to overlap.
extern "C" int ic_stream_create_cu_range(int first_cu, int n_cus, ic_stream_t* stream) {
    IC_CHECK_ARG(stream && first_cu >= 0 && n_cus > 0);
    int dev = 0, ncu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if (first_cu + n_cus > ncu || ncu > 1024) return IC_ERR_UNSUPPORTED;
    uint32_t words[32] = {0};
    for (int b = first_cu; b < first_cu + n_cus; ++b) words[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    e = hipExtStreamCreateWithCUMask(&s, (uint32_t)ic_cdiv(ncu, 32), words);
    if (e != hipSuccess) return (int)e;
    *stream = (ic_stream_t)s;
    return IC_OK;
}
extern "C" int ic_stream_destroy(ic_stream_t stream) { return stream ? (int)hipStreamDestroy((hipStream_t)stream) : IC_ERR_ARG; }

extern "C" int ic_event_create(void** ev) {
    IC_CHECK_ARG(ev);
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) return (int)r;
    *ev = (void*)e;
    return IC_OK;
}
extern "C" int ic_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : IC_ERR_ARG; }
extern "C" int ic_event_record(void* ev, ic_stream_t stream) {
    IC_CHECK_ARG(ev);
    return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}
extern "C" int ic_event_elapsed_ms(void* start, void* stop, float* ms) {
    IC_CHECK_ARG(start && stop && ms);
    hipError_t r = hipEventSynchronize((hipEvent_t)stop);
    if (r != hipSuccess) return (int)r;
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
