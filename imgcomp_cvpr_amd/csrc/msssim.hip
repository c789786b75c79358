// MS-SSIM training distortion and its gradient with respect to the reconstruction, as a handful of launches.
//
// Reference: code/ms_ssim.py:3-186 (the TF graph version used as the training loss), train.py:352-394 (Distortions:
// d_loss_scaled = K_ms_ssim * (1 - MS-SSIM(x, x_out))).  TensorFlow derives the gradient; the torch restatement
// (imgcomp_cvpr_amd/ms_ssim.py) is ~300 small kernels forward + backward -- 2 ms of a 17 ms training step at 15 % occupancy, and
// the HIP-graph replay that hid their launch cost turned out unreliable inside the training loop (training.py, GRAPH_LOSS).
// Here: one launch per scale forward, one per scale backward, a 2x2 box reduction between scales, one scalar kernel.
//
//   per scale l = 0..4 (images a_l = x, b_l = x_out, halved between scales by a 2x2 box after a REFLECT pad of (0,1)):
//     size = min(11, H_l, W_l); sigma = size * 1.5 / 11; Gaussian window of length 2 (size / 2) + 1, normalised; the image is
//     REFLECT-padded by (total_pad, total_pad / 2), total_pad = max(len - H_l, 0), when it is smaller than the window (:19-22);
//     mu, E[aa], E[bb], E[ab] by the separable 'VALID' blur; c1 = (0.01 * 255)^2, c2 = (0.03 * 255)^2;
//     cs = mean((2 cov + c2) / (var_a + var_b + c2)); ssim = mean(((2 mu_a mu_b + c1) (2 cov + c2)) / ((mu_a^2 + mu_b^2 + c1)(var_a + var_b + c2)))
//   MS-SSIM = ssim_4^w4 * prod_{l<4} cs_l^w_l, weights (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)            (:176-186)
//
// The blur of a scale is A_h . img . A_w^T with banded matrices A (row i holds the window at the padded positions it covers,
// reflected positions folded onto their source pixel): the forward kernel applies them as two passes through LDS, the backward
// kernel applies the transposes to the three per-position partials (d map / d mu_b, d E[bb], d E[ab]):
//     dL/db_l = s_l * (A^T P_mu A + 2 b . A^T P_bb A + a . A^T P_ab A) + box^T (dL/db_{l+1}),   s_l = dL/dS_l / positions_l.
// The matrices and their band limits depend on the shape only: ic_msssim_plan_fill() writes them into a HOST buffer once, the
// caller uploads it and passes the device copy with every call (no library state).  Means are reduced in float64 in a fixed
// order: results are bit-reproducible.
#include "common.h"
#include <math.h>
#include <string.h>

#define MS_LEVELS 5
#define MS_TR 16                   // tile rows (outputs forward, pixels backward)
#define MS_TC 64                   // tile columns
#define MS_RMAX (MS_TR + 10)       // rows of the intermediate a tile can need: window length <= 11
#define MS_MAXPART 4096            // partial sums per level the finishing kernel accepts per pass (it loops beyond)

static const double MS_WEIGHTS[MS_LEVELS] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};

struct MsLevel {
    int H, W, K, pb, oh, ow;       // image, window length, reflect pad before, blurred size
    // offsets in 4-byte words from the start of the plan blob
    int ah, aw;                    // A_h (oh x H), A_w (ow x W) floats
    int ylo, yhi, xlo, xhi;        // per output row / column: first and last input row / column of its window (ints, inclusive)
    int ilo, ihi, jlo, jhi;        // per input row / column: first and last output row / column whose window touches it
};
struct MsPlan {
    MsLevel lv[MS_LEVELS];
    int words;                     // size of the blob in 4-byte words
    int ok;
};

static int ms_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

static MsPlan ms_layout(int H, int W) {
    MsPlan p;
    memset(&p, 0, sizeof(p));
    int off = 0, h = H, w = W;
    p.ok = 1;
    for (int l = 0; l < MS_LEVELS; ++l) {
        MsLevel& v = p.lv[l];
        v.H = h; v.W = w;
        const int size = h < w ? (h < 11 ? h : 11) : (w < 11 ? w : 11);
        v.K = 2 * (size / 2) + 1;
        const int total = v.K - w > 0 ? v.K - w : 0;           // ms_ssim.py:19-22 on the NHWC tensor of :160-162: computed from the WIDTH, applied to both axes
        v.pb = total;
        const int pa = total / 2;
        v.oh = h + v.pb + pa - v.K + 1;
        v.ow = w + v.pb + pa - v.K + 1;
        // REFLECT needs pad < dimension; the 2x2 box behind every scale -- the reference builds it behind the last one too
        // (ms_ssim.py:171-175), so a 1-pixel last scale is an error there as well -- needs >= 2 pixels
        if (v.oh < 1 || v.ow < 1 || v.pb >= h || v.pb >= w || pa >= h || pa >= w || h < 2 || w < 2) p.ok = 0;
        if (!p.ok) break;
        v.ah = off; off += v.oh * h;
        v.aw = off; off += v.ow * w;
        v.ylo = off; off += v.oh;  v.yhi = off; off += v.oh;
        v.xlo = off; off += v.ow;  v.xhi = off; off += v.ow;
        v.ilo = off; off += h;     v.ihi = off; off += h;
        v.jlo = off; off += w;     v.jhi = off; off += w;
        h = (h + 1) / 2; w = (w + 1) / 2;
    }
    p.words = off;
    return p;
}

extern "C" size_t ic_msssim_plan_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const MsPlan p = ms_layout(H, W);
    return p.ok ? (size_t)p.words * 4 : 0;
}

static void ms_fill_axis(float* A, int* lo, int* hi, int* tlo, int* thi, int n_out, int n_in, int K, int pb, const double* k) {
    for (int y = 0; y < n_in; ++y) { tlo[y] = n_out; thi[y] = -1; }
    for (int i = 0; i < n_out; ++i) {
        float* row = A + (size_t)i * n_in;
        for (int y = 0; y < n_in; ++y) row[y] = 0.f;
        double acc[64];
        int mn = n_in, mx = -1;
        for (int y = 0; y < n_in && y < 64; ++y) acc[y] = 0.0;
        for (int u = 0; u < K; ++u) {
            const int y = ms_reflect(i + u - pb, n_in);
            if (n_in <= 64) acc[y] += k[u]; else row[y] += (float)k[u];     // reflections only occur on images smaller than the window
            if (y < mn) mn = y;
            if (y > mx) mx = y;
        }
        if (n_in <= 64) for (int y = mn; y <= mx; ++y) row[y] = (float)acc[y];
        lo[i] = mn; hi[i] = mx;
        for (int y = mn; y <= mx; ++y) if (row[y] != 0.f) { if (i < tlo[y]) tlo[y] = i; if (i > thi[y]) thi[y] = i; }
    }
}

// writes the plan blob of an H x W image into host memory (plain CPU arithmetic; call once per shape, upload, reuse)
extern "C" int ic_msssim_plan_fill(int H, int W, void* host_buf, size_t bytes) {
    IC_CHECK_ARG(host_buf && H > 0 && W > 0);
    const MsPlan p = ms_layout(H, W);
    if (!p.ok) return IC_ERR_UNSUPPORTED;
    IC_CHECK_ARG(bytes >= (size_t)p.words * 4);
    float* f = (float*)host_buf;
    int* n = (int*)host_buf;
    for (int l = 0; l < MS_LEVELS; ++l) {
        const MsLevel& v = p.lv[l];
        const int size = v.H < v.W ? (v.H < 11 ? v.H : 11) : (v.W < 11 ? v.W : 11);
        const double sigma = size * 1.5 / 11.0;
        double k[16], s = 0.0;
        const int half = size / 2;
        for (int u = 0; u < v.K; ++u) { const double x = u - half; k[u] = exp(-x * x / (2.0 * sigma * sigma)); s += fabs(k[u]); }
        for (int u = 0; u < v.K; ++u) k[u] /= s;
        ms_fill_axis(f + v.ah, n + v.ylo, n + v.yhi, n + v.ilo, n + v.ihi, v.oh, v.H, v.K, v.pb, k);
        ms_fill_axis(f + v.aw, n + v.xlo, n + v.xhi, n + v.jlo, n + v.jhi, v.ow, v.W, v.K, v.pb, k);
    }
    return IC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
struct MsFwdArgs {
    const float* a; const float* b;               // (planes, H, W)
    float* pmu; float* pbb; float* pab;           // (planes, oh, ow): d map / d mu_b, d E[bb], d E[ab]  (map = ssim when `last`, else cs)
    double* part;                                 // [work-group][2]: sum of cs, sum of ssim over the tile
    const float* Ah; const float* Aw;
    const int* ylo; const int* yhi; const int* xlo; const int* xhi;
    int H, W, oh, ow, tiles_y, tiles_x, last;
    float c1, c2;
};

__global__ __launch_bounds__(256) void msssim_level_fwd_kernel(const MsFwdArgs g) {
    __shared__ float hq[5][MS_RMAX][MS_TC + 1];
    __shared__ double red[2][4];
    const int t = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty = bid % g.tiles_y;
    const int plane = bid / g.tiles_y;
    const int i0 = ty * MS_TR, j0 = tx * MS_TC;
    const int ni = min(MS_TR, g.oh - i0), nj = min(MS_TC, g.ow - j0);
    const int y0 = g.ylo[i0], y1 = g.yhi[i0 + ni - 1];
    const int nr = y1 - y0 + 1;                                      // <= MS_RMAX (host-checked)
    const float* __restrict__ a = g.a + (size_t)plane * g.H * g.W;
    const float* __restrict__ b = g.b + (size_t)plane * g.H * g.W;
    // ---- pass 1: along the rows of the image: hq[q][r][j] = sum_x A_w[j][x] q(y0 + r, x) ----
    for (int e = t; e < nr * nj; e += 256) {
        const int r = e / nj, jj = e - r * nj, j = j0 + jj;
        const int x0 = g.xlo[j], x1 = g.xhi[j];
        const float* __restrict__ aw = g.Aw + (size_t)j * g.W;
        const float* __restrict__ ra = a + (size_t)(y0 + r) * g.W;
        const float* __restrict__ rb = b + (size_t)(y0 + r) * g.W;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
        for (int x = x0; x <= x1; ++x) {
            const float w = aw[x], va = ra[x], vb = rb[x];
            sa = fmaf(w, va, sa); sb = fmaf(w, vb, sb);
            saa = fmaf(w, va * va, saa); sbb = fmaf(w, vb * vb, sbb); sab = fmaf(w, va * vb, sab);
        }
        hq[0][r][jj] = sa; hq[1][r][jj] = sb; hq[2][r][jj] = saa; hq[3][r][jj] = sbb; hq[4][r][jj] = sab;
    }
    __syncthreads();
    // ---- pass 2: down the columns, then the per-position map and its partials ----
    double s_cs = 0.0, s_ssim = 0.0;
    for (int e = t; e < ni * nj; e += 256) {
        const int ii = e / nj, jj = e - ii * nj, i = i0 + ii, j = j0 + jj;
        const int ya = g.ylo[i], yb = g.yhi[i];
        const float* __restrict__ ah = g.Ah + (size_t)i * g.H;
        float ma = 0.f, mb = 0.f, eaa = 0.f, ebb = 0.f, eab = 0.f;
        for (int y = ya; y <= yb; ++y) {
            const float w = ah[y];
            const int r = y - y0;
            ma = fmaf(w, hq[0][r][jj], ma); mb = fmaf(w, hq[1][r][jj], mb);
            eaa = fmaf(w, hq[2][r][jj], eaa); ebb = fmaf(w, hq[3][r][jj], ebb); eab = fmaf(w, hq[4][r][jj], eab);
        }
        const float va = eaa - ma * ma, vb = ebb - mb * mb, cov = eab - ma * mb;
        const float v1 = 2.f * cov + g.c2, v2 = va + vb + g.c2;
        const float cs = v1 / v2;
        const float l1 = 2.f * ma * mb + g.c1, l2 = ma * ma + mb * mb + g.c1;
        const float lum = l1 / l2;
        s_cs += (double)cs;
        s_ssim += (double)(lum * cs);
        // d cs / d (E[ab], E[bb], mu_b)
        const float iv2 = 1.f / v2;
        float dab = 2.f * iv2;
        float dbb = -v1 * iv2 * iv2;
        float dmu = -2.f * ma * iv2 + 2.f * mb * v1 * iv2 * iv2;
        if (g.last) {                                               // ssim = lum * cs
            const float dl = 2.f * ma / l2 - l1 * 2.f * mb / (l2 * l2);
            dmu = cs * dl + lum * dmu;
            dab *= lum;
            dbb *= lum;
        }
        const size_t o = ((size_t)plane * g.oh + i) * g.ow + j;
        g.pmu[o] = dmu; g.pbb[o] = dbb; g.pab[o] = dab;
    }
    // ---- tile sums in float64, fixed order: lanes by xor tree, waves in order ----
    for (int m = 32; m >= 1; m >>= 1) {
        s_cs += __shfl_xor(s_cs, m);
        s_ssim += __shfl_xor(s_ssim, m);
    }
    if ((t & 63) == 0) { red[0][t >> 6] = s_cs; red[1][t >> 6] = s_ssim; }
    __syncthreads();
    if (t == 0) {
        g.part[2 * (size_t)blockIdx.x] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        g.part[2 * (size_t)blockIdx.x + 1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    }
}

// 2x2 box after a REFLECT pad of (0, 1), every second sample (ms_ssim.py:46-51): a and b of the next scale in one launch
__global__ __launch_bounds__(256) void msssim_halve_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ a2, float* __restrict__ b2,
                                                           int planes, int H, int W, int H2, int W2) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)planes * H2 * W2) return;
    const int j = (int)(idx % W2);
    const long long r = idx / W2;
    const int i = (int)(r % H2);
    const long long p = r / H2;
    const int y0 = 2 * i, y1 = (2 * i + 1 < H) ? 2 * i + 1 : H - 2;
    const int x0 = 2 * j, x1 = (2 * j + 1 < W) ? 2 * j + 1 : W - 2;
    const float* pa = a + p * (long long)H * W;
    const float* pb = b + p * (long long)H * W;
    // the separable form of the reference: rows first (0.5, 0.5), then columns
    a2[idx] = 0.5f * (0.5f * (pa[(size_t)y0 * W + x0] + pa[(size_t)y0 * W + x1]) + 0.5f * (pa[(size_t)y1 * W + x0] + pa[(size_t)y1 * W + x1]));
    b2[idx] = 0.5f * (0.5f * (pb[(size_t)y0 * W + x0] + pb[(size_t)y0 * W + x1]) + 0.5f * (pb[(size_t)y1 * W + x0] + pb[(size_t)y1 * W + x1]));
}

struct MsFinArgs {
    const double* part[MS_LEVELS];
    int nparts[MS_LEVELS];
    double count[MS_LEVELS];                  // positions the mean runs over: planes * oh * ow
    double weight[MS_LEVELS];
    float K;
    float* scalars;                           // [0] MS-SSIM, [1] K (1 - MS-SSIM), [2..6] cs_0..cs_3, ssim_4, [8..12] s_l = dL/dS_l / count_l
};

__global__ __launch_bounds__(256) void msssim_finish_kernel(const MsFinArgs g) {
    __shared__ double red[2][256];
    __shared__ double S[MS_LEVELS];
    const int t = threadIdx.x;
    for (int l = 0; l < MS_LEVELS; ++l) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = t; i < g.nparts[l]; i += 256) { s0 += g.part[l][2 * i]; s1 += g.part[l][2 * i + 1]; }
        red[0][t] = s0; red[1][t] = s1;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if (t < m) { red[0][t] += red[0][t + m]; red[1][t] += red[1][t + m]; }
            __syncthreads();
        }
        if (t == 0) S[l] = (l == MS_LEVELS - 1 ? red[1][0] : red[0][0]) / g.count[l];
        __syncthreads();
    }
    if (t == 0) {
        double ms = 1.0;
        for (int l = 0; l < MS_LEVELS; ++l) ms *= pow(S[l], g.weight[l]);        // a negative mean gives NaN, as tf.pow / torch do
        g.scalars[0] = (float)ms;
        g.scalars[1] = (float)((double)g.K * (1.0 - ms));
        g.scalars[7] = 0.f; g.scalars[13] = 0.f; g.scalars[14] = 0.f; g.scalars[15] = 0.f;
        for (int l = 0; l < MS_LEVELS; ++l) {
            g.scalars[2 + l] = (float)S[l];
            g.scalars[8 + l] = (float)(-(double)g.K * g.weight[l] * ms / S[l] / g.count[l]);
        }
    }
}

struct MsBwdArgs {
    const float* a; const float* b;               // (planes, H, W) of this scale
    const float* pmu; const float* pbb; const float* pab;       // (planes, oh, ow)
    const float* gnext;                           // dL/db of the next (coarser) scale (planes, H2, W2), or nullptr at the last scale
    float* gout;                                  // dL/db of this scale
    const float* scale;                           // device scalar s_l
    const float* Ah; const float* Aw;
    const int* ilo; const int* ihi; const int* jlo; const int* jhi;
    int H, W, oh, ow, H2, W2, tiles_y, tiles_x;
};

__global__ __launch_bounds__(256) void msssim_level_bwd_kernel(const MsBwdArgs g) {
    __shared__ float tq[3][MS_RMAX][MS_TC + 1];
    const int t = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty = bid % g.tiles_y;
    const int plane = bid / g.tiles_y;
    const int y0 = ty * MS_TR, x0 = tx * MS_TC;
    const int ny = min(MS_TR, g.H - y0), nx = min(MS_TC, g.W - x0);
    // output rows whose windows touch the tile's pixel rows (band limits are monotone in y)
    int ia = g.oh, ib = -1;
    for (int y = y0; y < y0 + ny; ++y) { ia = min(ia, g.ilo[y]); ib = max(ib, g.ihi[y]); }
    const int nr = ib - ia + 1;                                      // may be <= 0: no window touches these rows
    const float* __restrict__ pm = g.pmu + (size_t)plane * g.oh * g.ow;
    const float* __restrict__ pb = g.pbb + (size_t)plane * g.oh * g.ow;
    const float* __restrict__ pa = g.pab + (size_t)plane * g.oh * g.ow;
    // ---- pass 1: tq[q][r][x] = sum_j A_w[j][x] P_q(ia + r, j) ----
    for (int e = t; e < nr * nx; e += 256) {
        const int r = e / nx, xx = e - r * nx, x = x0 + xx;
        const int ja = g.jlo[x], jb = g.jhi[x];
        const size_t row = (size_t)(ia + r) * g.ow;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int j = ja; j <= jb; ++j) {
            const float w = g.Aw[(size_t)j * g.W + x];
            s0 = fmaf(w, pm[row + j], s0); s1 = fmaf(w, pb[row + j], s1); s2 = fmaf(w, pa[row + j], s2);
        }
        tq[0][r][xx] = s0; tq[1][r][xx] = s1; tq[2][r][xx] = s2;
    }
    __syncthreads();
    const float sc = *g.scale;
    for (int e = t; e < ny * nx; e += 256) {
        const int yy = e / nx, xx = e - yy * nx, y = y0 + yy, x = x0 + xx;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int i = g.ilo[y]; i <= g.ihi[y]; ++i) {
            const float w = g.Ah[(size_t)i * g.H + y];
            const int r = i - ia;
            s0 = fmaf(w, tq[0][r][xx], s0); s1 = fmaf(w, tq[1][r][xx], s1); s2 = fmaf(w, tq[2][r][xx], s2);
        }
        const size_t o = ((size_t)plane * g.H + y) * g.W + x;
        float v = sc * (s0 + 2.f * g.b[o] * s1 + g.a[o] * s2);
        if (g.gnext) {
            // transpose of the 2x2 box behind the (0,1) REFLECT pad: pixel y feeds output row y / 2, and on an odd-sized axis
            // the pixel before the last is also the reflection read by the last output row
            const float* gn = g.gnext + (size_t)plane * g.H2 * g.W2;
            const int i0 = y >> 1, i1 = ((g.H & 1) && y == g.H - 2) ? g.H2 - 1 : -1;
            const int j0 = x >> 1, j1 = ((g.W & 1) && x == g.W - 2) ? g.W2 - 1 : -1;
            float u = gn[(size_t)i0 * g.W2 + j0];
            if (j1 >= 0) u += gn[(size_t)i0 * g.W2 + j1];
            if (i1 >= 0) { u += gn[(size_t)i1 * g.W2 + j0]; if (j1 >= 0) u += gn[(size_t)i1 * g.W2 + j1]; }
            v = fmaf(0.25f, u, v);
        }
        g.gout[o] = v;
    }
}

// workspace: pyramids of a and b (scales 1..4), the three partial maps of every scale, the gradients of scales 1..4, partial sums
struct MsWs { size_t a[MS_LEVELS], b[MS_LEVELS], p[MS_LEVELS], gr[MS_LEVELS], part[MS_LEVELS], total; int nparts[MS_LEVELS]; };

static MsWs ms_workspace(const MsPlan& p, size_t planes) {
    MsWs w;
    memset(&w, 0, sizeof(w));
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    for (int l = 0; l < MS_LEVELS; ++l) {
        const MsLevel& v = p.lv[l];
        if (l) { w.a[l] = take(planes * v.H * v.W * 4); w.b[l] = take(planes * v.H * v.W * 4); w.gr[l] = take(planes * v.H * v.W * 4); }
        w.p[l] = take(3 * planes * v.oh * v.ow * 4);
        w.nparts[l] = (int)(planes * ic_cdiv(v.oh, MS_TR) * ic_cdiv(v.ow, MS_TC));
        w.part[l] = take((size_t)w.nparts[l] * 16);
    }
    w.total = off;
    return w;
}

extern "C" size_t ic_msssim_workspace_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const MsPlan p = ms_layout(H, W);
    if (!p.ok) return 0;
    return ms_workspace(p, (size_t)N * C).total;
}

// x, x_out: (N, C, H, W) float32, values 0..max_val = 255.  plan_dev: the device copy of ic_msssim_plan_fill(H, W).
// grad_out (N, C, H, W): d (K (1 - MS-SSIM)) / d x_out.  scalars_out: 16 device floats -- [0] MS-SSIM, [1] K (1 - MS-SSIM),
// [2..5] cs of scales 0..3, [6] ssim of scale 4, [8..12] the per-position gradient scales.  grad_out may be NULL (value only).
extern "C" int ic_msssim_loss_grad_f32(const float* x, const float* x_out, int N, int C, int H, int W, float K, const void* plan_dev,
                                       float* grad_out, float* scalars_out, void* workspace, size_t workspace_bytes, ic_stream_t stream) {
    IC_CHECK_ARG(x && x_out && plan_dev && scalars_out && workspace && N > 0 && C > 0 && H > 0 && W > 0);
    const MsPlan p = ms_layout(H, W);
    if (!p.ok) return IC_ERR_UNSUPPORTED;
    const size_t planes = (size_t)N * C;
    const MsWs w = ms_workspace(p, planes);
    IC_CHECK_ARG(workspace_bytes >= w.total);
    if (planes * (size_t)H * W >= ((size_t)1 << 31)) return IC_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const float* pf = (const float*)plan_dev;
    const int* pi = (const int*)plan_dev;
    const float* a[MS_LEVELS]; const float* b[MS_LEVELS];
    a[0] = x; b[0] = x_out;
    const float c1 = (0.01f * 255.f) * (0.01f * 255.f), c2 = (0.03f * 255.f) * (0.03f * 255.f);
    MsFinArgs fin;
    memset(&fin, 0, sizeof(fin));
    for (int l = 0; l < MS_LEVELS; ++l) {
        const MsLevel& v = p.lv[l];
        if (l) {
            const MsLevel& u = p.lv[l - 1];
            float* a2 = (float*)(ws + w.a[l]); float* b2 = (float*)(ws + w.b[l]);
            const long long n = (long long)planes * v.H * v.W;
            hipLaunchKernelGGL(msssim_halve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a[l - 1], b[l - 1], a2, b2,
                               (int)planes, u.H, u.W, v.H, v.W);
            a[l] = a2; b[l] = b2;
        }
        MsFwdArgs f;
        f.a = a[l]; f.b = b[l];
        float* P = (float*)(ws + w.p[l]);
        const size_t pn = planes * v.oh * v.ow;
        f.pmu = P; f.pbb = P + pn; f.pab = P + 2 * pn;
        f.part = (double*)(ws + w.part[l]);
        f.Ah = pf + v.ah; f.Aw = pf + v.aw;
        f.ylo = pi + v.ylo; f.yhi = pi + v.yhi; f.xlo = pi + v.xlo; f.xhi = pi + v.xhi;
        f.H = v.H; f.W = v.W; f.oh = v.oh; f.ow = v.ow;
        f.tiles_y = ic_cdiv(v.oh, MS_TR); f.tiles_x = ic_cdiv(v.ow, MS_TC);
        f.last = l == MS_LEVELS - 1;
        f.c1 = c1; f.c2 = c2;
        hipLaunchKernelGGL(msssim_level_fwd_kernel, dim3((unsigned)w.nparts[l]), dim3(256), 0, st, f);
        fin.part[l] = f.part; fin.nparts[l] = w.nparts[l];
        fin.count[l] = (double)planes * v.oh * v.ow;
        fin.weight[l] = MS_WEIGHTS[l];
    }
    fin.K = K; fin.scalars = scalars_out;
    hipLaunchKernelGGL(msssim_finish_kernel, dim3(1), dim3(256), 0, st, fin);
    if (grad_out) {
        for (int l = MS_LEVELS - 1; l >= 0; --l) {
            const MsLevel& v = p.lv[l];
            MsBwdArgs g;
            g.a = a[l]; g.b = b[l];
            const float* P = (const float*)(ws + w.p[l]);
            const size_t pn = planes * v.oh * v.ow;
            g.pmu = P; g.pbb = P + pn; g.pab = P + 2 * pn;
            g.gnext = l + 1 < MS_LEVELS ? (const float*)(ws + w.gr[l + 1]) : nullptr;
            g.gout = l ? (float*)(ws + w.gr[l]) : grad_out;
            g.scale = scalars_out + 8 + l;
            g.Ah = pf + v.ah; g.Aw = pf + v.aw;
            g.ilo = pi + v.ilo; g.ihi = pi + v.ihi; g.jlo = pi + v.jlo; g.jhi = pi + v.jhi;
            g.H = v.H; g.W = v.W; g.oh = v.oh; g.ow = v.ow;
            g.H2 = l + 1 < MS_LEVELS ? p.lv[l + 1].H : 0; g.W2 = l + 1 < MS_LEVELS ? p.lv[l + 1].W : 0;
            g.tiles_y = ic_cdiv(v.H, MS_TR); g.tiles_x = ic_cdiv(v.W, MS_TC);
            hipLaunchKernelGGL(msssim_level_bwd_kernel, dim3((unsigned)(planes * g.tiles_y * g.tiles_x)), dim3(256), 0, st, g);
        }
    }
    IC_LAUNCH_CHECK();
    return IC_OK;
}
