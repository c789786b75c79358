// Probe: which physical CUs does a hipExtStreamCreateWithCUMask stream use?  Prints, per mask, the number of distinct
// (XCC, SE, CU) a launch of 2048 spinning work-groups touched and the per-XCC CU counts.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o ab/cumask_probe   (ab/ is git-ignored scratch that still travels to the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <tuple>

__global__ void probe(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the CU busy so that the dispatcher has to spread the work-groups
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc | (v == 12345.f ? 1u << 31 : 0u); }
}

static void run(const char* name, const std::vector<int>& bits) {
    unsigned words[8] = {0};
    for (int b : bits) words[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, words) != hipSuccess) { printf("%s: create failed\n", name); return; }
    unsigned back[8] = {0};
    hipExtStreamGetCUMask(s, 8, back);
    const int NB = 2048;
    unsigned* d; hipMalloc(&d, NB * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(NB), dim3(256), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(NB * 2);
    hipMemcpy(h.data(), d, NB * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<std::tuple<int, int, int, int>> cus;
    int per_xcc[8] = {0};
    for (int i = 0; i < NB; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        if (cus.insert({(int)xcc, se, sh, cu}).second) per_xcc[xcc & 7]++;
    }
    printf("%-28s bits %3zu  mask back %08x %08x .. %08x  distinct CUs %3zu  per XCC:", name, bits.size(), back[0], back[1], back[7], cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    hipFree(d); hipStreamDestroy(s);
}

int main() {
    std::vector<int> all, first192, last64, first128, blk;
    for (int i = 0; i < 256; ++i) all.push_back(i);
    for (int i = 0; i < 192; ++i) first192.push_back(i);
    for (int i = 192; i < 256; ++i) last64.push_back(i);
    for (int i = 0; i < 128; ++i) first128.push_back(i);
    for (int x = 0; x < 8; ++x) for (int i = 0; i < 24; ++i) blk.push_back(32 * x + i);
    run("all 256", all);
    run("bits 0..191", first192);
    run("bits 192..255", last64);
    run("bits 0..127", first128);
    run("24 of every 32", blk);
    std::vector<int> one = {0}, eight = {0, 1, 2, 3, 4, 5, 6, 7};
    run("bit 0", one);
    run("bits 0..7", eight);
    return 0;
}
