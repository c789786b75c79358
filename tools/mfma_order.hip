// Which scalar expression reproduces v_mfma_f32_32x32x2_f32 bit for bit?  (Needed by the sequential decoder's cached form:
// one new voxel per layer per symbol is computed on the vector units and must equal the matrix-core result of the parallel pass.)
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_order.hip -o tools/mfma_order.bin; gpurun -- ./tools/mfma_order.bin   (the binary is git-ignored)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// chain of NS MFMAs: acc += A_s (32 x 2) * B_s (2 x 32); lane l supplies A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31]
__global__ void k_mfma(const float* A, const float* B, const float* C, float* D, int NS) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[r * 64 + lane];
    for (int s = 0; s < NS; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s * 64 + lane], B[s * 64 + lane], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[r * 64 + lane] = acc[r];
}
__global__ void k_mfma16(const float* A, const float* B, const float* C, float* D, int NS) {   // 16x16x4: lane: A[m = l & 15][k = l >> 4]
    const int lane = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[r * 64 + lane];
    for (int s = 0; s < NS; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s * 64 + lane], B[s * 64 + lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[r * 64 + lane] = acc[r];
}

int main() {
    const int NS = 42;
    float *hA = new float[NS * 64], *hB = new float[NS * 64], *hC = new float[16 * 64], *hD = new float[16 * 64], *hD16 = new float[4 * 64];
    srand(7);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX) * 4.0 - 2.0) * (rand() % 7 == 0 ? 1e-3f : 1.f); };
    for (int i = 0; i < NS * 64; ++i) { hA[i] = rnd(); hB[i] = rnd(); }
    for (int i = 0; i < 16 * 64; ++i) hC[i] = rnd();
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, NS * 64 * 4); hipMalloc(&dB, NS * 64 * 4); hipMalloc(&dC, 16 * 64 * 4); hipMalloc(&dD, 16 * 64 * 4);
    hipMemcpy(dA, hA, NS * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, NS * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, 16 * 64 * 4, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dC, dD, NS);
    hipMemcpy(hD, dD, 16 * 64 * 4, hipMemcpyDeviceToHost);
    k_mfma16<<<1, 64>>>(dA, dB, dC, dD, NS);
    hipMemcpy(hD16, dD, 4 * 64 * 4, hipMemcpyDeviceToHost);
    // 32x32x2: output register r of lane l = D[m = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][n = l & 31]
    const char* names[4] = {"fma chain k = 0 then 1", "fma chain k = 1 then 0", "c + (a0 b0 + a1 b1) products in double", "separate mul + add, k ascending"};
    for (int cand = 0; cand < 4; ++cand) {
        int bad = 0;
        for (int r = 0; r < 16; ++r) for (int l = 0; l < 64; ++l) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
            float acc = hC[r * 64 + l];
            for (int s = 0; s < NS; ++s) {
                const float a0 = hA[s * 64 + m], a1 = hA[s * 64 + 32 + m], b0 = hB[s * 64 + n], b1 = hB[s * 64 + 32 + n];
                if (cand == 0) acc = fmaf(a1, b1, fmaf(a0, b0, acc));
                else if (cand == 1) acc = fmaf(a0, b0, fmaf(a1, b1, acc));
                else if (cand == 2) acc = (float)((double)acc + ((double)a0 * b0 + (double)a1 * b1));
                else { volatile float p0 = a0 * b0; volatile float t = acc + p0; volatile float p1 = a1 * b1; acc = t + p1; }
            }
            if (memcmp(&acc, &hD[r * 64 + l], 4) != 0) ++bad;
        }
        printf("32x32x2  %-42s mismatches %d / 1024\n", names[cand], bad);
    }
    // 16x16x4: register r of lane l = D[m = 4 (l >> 4) + r][n = l & 15]; A[m][k]: lane (k * 16 + m)
    for (int cand = 0; cand < 2; ++cand) {
        int bad = 0;
        for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) {
            const int m = 4 * (l >> 4) + r, n = l & 15;
            float acc = hC[r * 64 + l];
            for (int s = 0; s < NS; ++s)
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = cand == 0 ? kk : 3 - kk;
                    acc = fmaf(hA[s * 64 + 16 * k + m], hB[s * 64 + 16 * k + n], acc);
                }
            if (memcmp(&acc, &hD16[r * 64 + l], 4) != 0) ++bad;
        }
        printf("16x16x4  fma chain k %s                      mismatches %d / 256\n", cand == 0 ? "ascending " : "descending", bad);
    }
    return 0;
}
