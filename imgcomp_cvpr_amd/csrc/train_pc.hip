// Backward pieces of the context model (reference: gradients of probclass.py:63-106,185-261 as tf.gradients builds
// them for the Adam_PC optimiser, train.py:339-349).  The filter gradients are in conv_wgrad.hip (3-D mode);
// here: the cross-entropy/ReLU gradient of the logits, the data gradient of a masked VALID conv3d, channel sums.
// The context model is ~2 % of a training step's FLOPs: these are plain one-lane-per-output kernels.
#include "common.h"
#include "internal.h"

// g[n][l][v] = (softmax(logits[n][v])[l] - [l == sym]) * log2(e) * d_bits[n][v] * [logits[n][v][l] > 0]
// logits: (N, vol, L) channels-last, post-ReLU (probclass.py:220,233); g: (N, L, vol) planar.
__global__ __launch_bounds__(256) void pc_dlogits_kernel(const float* __restrict__ logits, const long long* __restrict__ sym,
                                                         const float* __restrict__ dbits, float* __restrict__ g,
                                                         int N, int vol, int L) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * vol) return;
    const int n = (int)(i / vol), v = (int)(i - (long long)n * vol);
    const float* l = logits + i * L;
    float m = l[0];
    for (int j = 1; j < L; ++j) m = fmaxf(m, l[j]);
    float e[16], s = 0.f;
    for (int j = 0; j < L; ++j) { e[j] = expf(l[j] - m); s += e[j]; }
    const float up = dbits[i] * 1.44269504f;
    const int sy = (int)sym[i];
    for (int j = 0; j < L; ++j) {
        float gv = (e[j] / s - (j == sy ? 1.f : 0.f)) * up;
        if (!(l[j] > 0.f)) gv = 0.f;
        g[((size_t)n * L + j) * vol + v] = gv;
    }
}

// dx[n][ci][u] = ( sum_{live taps t, co} g[n][co][u - off(t)] * w[t][ci][co]  (+ res[n][ci][u - 2] when inside) ) * [act > 0]
//   g: (N,Cout,OD,OH,OW)   dx/act: (N,Cin,OD+1,OH+2,OW+2)   res: (N,Cin,OD-1,OH-2,OW-2) embedded at offset (2,2,2)
struct PcBwdArgs {
    const float* g; const float* w; const float* res; const float* act; float* dx;
    int N, Cin, Cout, OD, OH, OW, first_mask, relu_mask;
};

template <int CIB>
__global__ __launch_bounds__(256) void pc_bwd_data_kernel(const PcBwdArgs a) {
    const int n = blockIdx.z, ci0 = blockIdx.y * CIB;
    const int D = a.OD + 1, H = a.OH + 2, W = a.OW + 2;
    const int ivol = D * H * W, ovol = a.OD * a.OH * a.OW;
    int u = blockIdx.x * 256 + threadIdx.x;
    const bool live = u < ivol;
    if (!live) u = ivol - 1;
    const int x = u % W, t2 = u / W, y = t2 % H, d = t2 / H;
    float acc[CIB];
#pragma unroll
    for (int j = 0; j < CIB; ++j) acc[j] = 0.f;
    int cofs[CIB];
#pragma unroll
    for (int j = 0; j < CIB; ++j) cofs[j] = min(ci0 + j, a.Cin - 1) * a.Cout;
    const float* gn = a.g + (size_t)n * a.Cout * ovol;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool dead = (kd == 1) && (kh == 2 || (kh == 1 && (a.first_mask ? kw >= 1 : kw >= 2)));
                if (dead) continue;
                const int od = d - kd, oy = y - kh, ox = x - kw;
                const bool in = od >= 0 && od < a.OD && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
                const int go = in ? (od * a.OH + oy) * a.OW + ox : 0;
                const float* wt = a.w + (size_t)((kd * 3 + kh) * 3 + kw) * a.Cin * a.Cout;
                for (int co = 0; co < a.Cout; ++co) {
                    const float gv = in ? gn[(size_t)co * ovol + go] : 0.f;
#pragma unroll
                    for (int j = 0; j < CIB; ++j) acc[j] = fmaf(gv, wt[cofs[j] + co], acc[j]);
                }
            }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < CIB; ++j) {
        const int ci = ci0 + j;
        if (ci >= a.Cin) break;
        float v = acc[j];
        if (a.res) {
            const int rd = d - 2, ry = y - 2, rx = x - 2;
            const int RD = a.OD - 1, RH = a.OH - 2, RW = a.OW - 2;
            if (rd >= 0 && rd < RD && ry >= 0 && ry < RH && rx >= 0 && rx < RW)
                v += a.res[(((size_t)n * a.Cin + ci) * RD + rd) * RH * RW + (size_t)ry * RW + rx];
        }
        const size_t o = ((size_t)n * a.Cin + ci) * ivol + u;
        if (a.relu_mask && !(a.act[o] > 0.f)) v = 0.f;
        a.dx[o] = v;
    }
}

extern "C" int ic_pc_dlogits_f32(const float* logits, const int64_t* symbols, const float* d_bits, float* g,
                                 int N, int vol, int L, ic_stream_t stream) {
    IC_CHECK_ARG(logits && symbols && d_bits && g && N > 0 && vol > 0 && L > 0);
    if (L > 16) return IC_ERR_UNSUPPORTED;
    const long long total = (long long)N * vol;
    hipLaunchKernelGGL(pc_dlogits_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       (const long long*)symbols, d_bits, g, N, vol, L);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// epilogue of the matrix-core path: dx = (raw (+ res inside)) * [act > 0], in place
__global__ __launch_bounds__(256) void pc_bwd_fix_kernel(const PcBwdArgs a, long long total) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int D = a.OD + 1, H = a.OH + 2, W = a.OW + 2;
    const int x = (int)(o % W);
    long long r = o / W;
    const int y = (int)(r % H); r /= H;
    const int d = (int)(r % D); r /= D;           // r = n * Cin + ci
    float v = a.dx[o];
    if (a.res) {
        const int rd = d - 2, ry = y - 2, rx = x - 2;
        const int RD = a.OD - 1, RH = a.OH - 2, RW = a.OW - 2;
        if (rd >= 0 && rd < RD && ry >= 0 && ry < RH && rx >= 0 && rx < RW)
            v += a.res[((size_t)r * RD + rd) * RH * RW + (size_t)ry * RW + rx];
    }
    if (a.relu_mask && !(a.act[o] > 0.f)) v = 0.f;
    a.dx[o] = v;
}

extern "C" size_t ic_pc_bwd_data_workspace_bytes(int N, int Cin, int Cout, int OD, int OH, int OW) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || OD <= 0 || OH <= 0 || OW <= 0) return 0;
    const size_t b = icx_pc_bwd_data_mfma_workspace(N, Cin, Cout, OD, OH, OW);
    return b ? b + 64 * sizeof(float) : 0;
}

// workspace: ic_pc_bwd_data_workspace_bytes(...) bytes select the matrix-core path ("other" mask, Cin = 24 or 64); NULL or
// 0 bytes (or a shape it does not cover) run the any-shape VALU kernel.
extern "C" int ic_pc_bwd_data_f32(const float* g, const float* w, const float* res, const float* act, float* dx,
                                  int N, int Cin, int Cout, int OD, int OH, int OW, int first_mask, int relu_mask,
                                  void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(g && w && dx && N > 0 && Cin > 0 && Cout > 0 && OD > 0 && OH > 0 && OW > 0);
    IC_CHECK_ARG(!relu_mask || act);
    PcBwdArgs a{};
    a.g = g; a.w = w; a.res = res; a.act = act; a.dx = dx;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.OD = OD; a.OH = OH; a.OW = OW; a.first_mask = first_mask; a.relu_mask = relu_mask;
    const int ivol = (OD + 1) * (OH + 2) * (OW + 2);
    hipStream_t st = (hipStream_t)stream;
    const size_t need = first_mask ? 0 : ic_pc_bwd_data_workspace_bytes(N, Cin, Cout, OD, OH, OW);
    if (workspace && need && workspace_bytes >= need) {
        float* zero = (float*)((char*)workspace + need - 64 * sizeof(float));
        if (hipMemsetAsync(zero, 0, 64 * sizeof(float), st) != hipSuccess) return IC_ERR_ARG;
        int rc = icx_pc_bwd_data_mfma(g, w, dx, N, Cin, Cout, OD, OH, OW, zero, workspace, need - 64 * sizeof(float), st);
        if (rc) return rc;
        if (res || relu_mask) {
            const long long total = (long long)N * Cin * ivol;
            hipLaunchKernelGGL(pc_bwd_fix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, total);
        }
        IC_LAUNCH_CHECK();
        return IC_OK;
    }
    hipLaunchKernelGGL((pc_bwd_data_kernel<8>), dim3(ic_cdiv(ivol, 256), ic_cdiv(Cin, 8), N), dim3(256), 0, st, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- per-channel sums of an (N, C, M) tensor (bias gradients): float64 two-stage, deterministic ----
#define CS_CHUNKS 32
__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ x, int N, int C, int M,
                                                          double* __restrict__ partial) {
    const int c = blockIdx.x, chunk = blockIdx.y;
    const long long T = (long long)N * M, per = (T + CS_CHUNKS - 1) / CS_CHUNKS;
    const long long lo = per * chunk, hi = lo + per < T ? lo + per : T;
    double s = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const long long n = i / M, p = i - n * M;
        s += x[((size_t)n * C + c) * M + p];
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)c * CS_CHUNKS + chunk] = sh[0];
}
__global__ void channel_sum_stage2(const double* __restrict__ partial, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < CS_CHUNKS; ++k) s += partial[(size_t)c * CS_CHUNKS + k];
    out[c] = (float)s;
}

extern "C" size_t ic_channel_sum_workspace_bytes(int C) { return C > 0 ? (size_t)C * CS_CHUNKS * sizeof(double) : 0; }

extern "C" int ic_channel_sum_f32(const float* x, float* out, int N, int C, int M, void* workspace, ic_stream_t stream) {
    IC_CHECK_ARG(x && out && workspace && N > 0 && C > 0 && M > 0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_sum_stage1, dim3(C, CS_CHUNKS), dim3(256), 0, st, x, N, C, M, (double*)workspace);
    hipLaunchKernelGGL(channel_sum_stage2, dim3(ic_cdiv(C, 64)), dim3(64), 0, st, (const double*)workspace, C, out);
    IC_LAUNCH_CHECK();
    return IC_OK;
}

// ---- tf.train.AdamOptimizer on a flat bucket (train.py:339-349 via training_helpers.py:38-48) ------------------------------
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  var -= lr_t m / (sqrt(v) + eps)      (epsilon OUTSIDE the bias correction;
//   lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) comes from the host).  One pass over four arrays instead of seven multi-tensor
//   passes: parameters, gradients and both slots of a variable group share one flat layout.
__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long long n, float lr_t, float b1, float b2, float eps) {
    const float c1 = 1.f - b1, c2 = 1.f - b2;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ma[j] = fmaf(c1, ga[j], ma[j] * b1);
            va[j] = fmaf(c2 * ga[j], ga[j], va[j] * b2);
            pa[j] = fmaf(-lr_t, ma[j] / (sqrtf(va[j]) + eps), pa[j]);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = fmaf(c1, gi, m[i] * b1), vi = fmaf(c2 * gi, gi, v[i] * b2);
        m[i] = mi; v[i] = vi;
        p[i] = fmaf(-lr_t, mi / (sqrtf(vi) + eps), p[i]);
    }
}

extern "C" int ic_adam_tf_f32(float* var, const float* grad, float* m, float* v, long long count, float lr_t, float beta1,
                              float beta2, float eps, ic_stream_t stream) {
    IC_CHECK_ARG(var && grad && m && v && count > 0);
    if ((((size_t)var | (size_t)grad | (size_t)m | (size_t)v) & 15) != 0) return IC_ERR_ARG;      // 16-byte accesses
    const long long blocks = (count / 4 + 255) / 256;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0,
                       (hipStream_t)stream, var, grad, m, v, count, lr_t, beta1, beta2, eps);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
