// Do two work-groups with 72 KB of static LDS each really get disjoint LDS on a gfx950 CU?  (round 4, F(4x4) kernel debug)
// Every work-group fills its LDS with its id, spins ~20 us so that a second work-group is resident beside it, then verifies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define WORDS (73728 / 4)
__global__ __launch_bounds__(256) void probe(int* bad, int spin) {
    __shared__ int lds[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) lds[i] = blockIdx.x * 100000 + i;
    __syncthreads();
    long long t0 = clock64();
    const long long mine = spin + (long long)((blockIdx.x * 2654435761u) >> 20) * 16;      // 0 .. 65 k clocks extra: work-groups finish out of step
    while (clock64() - t0 < mine) {}
    // touch the LDS while waiting for nothing: re-verify half-way through as well
    __syncthreads();
    __syncthreads();
    int b = 0;
    for (int i = threadIdx.x; i < WORDS; i += 256) b += lds[i] != (int)(blockIdx.x * 100000 + i);
    if (b) atomicAdd(bad, b);
}
int main() {
    int* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe, dim3(8192), dim3(256), 0, 0, bad, 50000);
    int h = -1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, 0);
    printf("mismatching words: %d   (occupancy query: %d work-groups per CU)\n", h, occ);
    return 0;
}
