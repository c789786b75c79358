// Does a wave block at v_mfma issue until the matrix pipe takes the instruction, or can it run ahead of queued MFMAs?
// (round 4: the F(4x4) kernel read stale accumulators right behind its loop with two waves per SIMD)
// One work-group of `threads` threads; every wave issues N independent 16x16x4 fp32 MFMAs and stamps s_memtime before / after
// the ISSUE of the batch, then pads and reads the results.  With a blocking issue, issue time ~ N x 32 clocks x (waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int N>
__global__ void probe(unsigned long long* out, float* sink) {
    f4 acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i])); }
    float a = 1.0f + threadIdx.x, b = 2.0f;
    asm volatile("" : "+v"(a), "+v"(b));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) { asm volatile("" : "+a"(acc[i])); s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { out[(threadIdx.x >> 6) * 2] = t1 - t0; out[(threadIdx.x >> 6) * 2 + 1] = t2 - t1; }
    sink[threadIdx.x] = s;
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 16 * 16); hipMalloc(&sink, 4096);
    for (int threads : {64, 256, 512}) {
        hipMemset(out, 0, 256);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<32>, dim3(1), dim3(threads), 0, 0, out, sink);
        unsigned long long h[32]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
        printf("%3d threads (%d waves, %d per SIMD): 128 MFMAs issued in %llu clocks (s_memtime ticks = 100 MHz x ? -- see ratio), read-out %llu; wave 1: %llu\n",
               threads, threads / 64, threads > 256 ? 2 : 1, h[0], h[1], h[2]);
    }
    return 0;
}
