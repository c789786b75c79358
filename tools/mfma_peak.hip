// Calibration micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on THIS box, alone and when its
// B operand is fed from LDS by ds_read_b32 / b64 / b128?
//   build: hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: operands in registers; 1: ds_read_b32 per MFMA; 2: ds_read_b64 per 2 MFMAs; 3: ds_read_b128 per 4
// MFMAs; 4: registers + one extra VALU op per MFMA.  All LDS reads are conflict-free, immediate offsets,
// issued one group (NACC*4 MFMAs) ahead of use (register double buffer).
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = in[tid], bv = in[tid + 256];
    const int lane = tid & 63;
    const float* L1 = lds + lane;            // b32: lane-consecutive dwords
    const float* L2 = lds + 2 * lane;        // b64
    const float* L4 = lds + 4 * lane;        // b128
    float bq[2][4][NACC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < NACC; ++a) bq[0][k][a] = bv + k + a;
    float extra = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int cur = g & 1;
            // prefetch next group's B operands
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int o = ((it & 3) * 2 + g) * 256 * NACC + a * 256;
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) bq[cur ^ 1][k][a] = L1[o + 64 * k];
                } else if (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const f32x2 v = *reinterpret_cast<const f32x2*>(L2 + o + 64 * k);
                        bq[cur ^ 1][k][a] = v[0]; bq[cur ^ 1][k + 1][a] = v[1];
                    }
                } else if (MODE == 3) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(L4 + o);
#pragma unroll
                    for (int k = 0; k < 4; ++k) bq[cur ^ 1][k][a] = v[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) bq[cur ^ 1][k][a] = bq[cur][k][a];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < NACC; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[cur][k][a], acc[a], 0, 0, 0);
                    if (MODE == 4) extra = __builtin_fmaf(extra, 1.0001f, av);
                }
            if (MODE >= 1 && MODE <= 3) {
#pragma unroll
                for (int i = 0; i < 4 * NACC; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = extra;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int NACC, int MODE>
void run(const char* name, int blocks, const float* din, float* dout, unsigned long long* dclk) {
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(blocks), dim3(256), 0, 0, din, dout, iters, dclk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    unsigned long long clk; hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NACC;                 // MFMAs per wave
    const double flop = nm * 4 * blocks * 2.0 * 32 * 32 * 2;
    printf("%-34s blocks=%4d  %8.1f us  %7.1f TFLOP/s  shader clocks/MFMA (wave 0) %.1f\n",
           name, blocks, ms * 1e3, flop / (ms * 1e-3) / 1e12, clk / nm);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    std::vector<float> h(8192);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    float *din, *dout; unsigned long long* dclk;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 1 << 22); hipMalloc(&dclk, 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<3, 0>("NACC=3 registers only", blocks, din, dout, dclk);
        run<3, 4>("NACC=3 registers + 1 VALU/MFMA", blocks, din, dout, dclk);
        run<3, 1>("NACC=3 ds_read_b32 per MFMA", blocks, din, dout, dclk);
        run<3, 2>("NACC=3 ds_read_b64 per 2 MFMA", blocks, din, dout, dclk);
        run<3, 3>("NACC=3 ds_read_b128 per 4 MFMA", blocks, din, dout, dclk);
        run<4, 3>("NACC=4 ds_read_b128 per 4 MFMA", blocks, din, dout, dclk);
    }
    return 0;
}
