// Cross-replica sums without a collective library call: every rank PUSHES its contribution into a slot of every peer's
// exchange region (peer-mapped device memory: xGMI stores between GPUs), raises a flag there, waits for the flags in its OWN
// region and adds the contributions up in rank order.  One 256-thread work-group, one launch per exchange.
//
// Why: training-mode BatchNorm over a batch that is split across the GPUs (autoencoder.py:106-125 normalises over the whole
// batch of one device; train.py:150-153 feeds one batch) needs 2 C float64 moments summed over the ranks once per layer and
// direction -- 140 exchanges of <= 2 KB per training step, each on the critical path.  An RCCL all-reduce of that size is
// launch- and protocol-latency bound (tens of microseconds); this kernel is one store + one flag + one poll per peer.
// The sum order is r = 0 .. W-1 on every rank: all ranks hold bit-identical results, whatever arrives first.
//
// Memory: the region is fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)): not cached
// incoherently by the owner's L2, so a peer's stores are visible to a running kernel.  Payload and flags are written and read
// with system-scope accesses; the flag follows the payload behind a system-scope release fence (+ an explicit vmcnt(0):
// MI355X_MICROARCH.md, "Compiler hazard").  Slots rotate (PX_SLOTS = 4): a rank can run at most one exchange ahead of the
// slowest one, so a slot is rewritten only after every reader has left it.  Flags carry the exchange's sequence number
// (monotonic, never reset).  Every spin is bounded; a time-out sets *status and leaves the local values unchanged.
#include "common.h"
#include <string.h>

#define PX_MAX_WORLD 8
#define PX_SLOTS 4
#define PX_MAXN 1024                              // doubles per contribution
#define PX_FLAG_STRIDE 16                         // one flag per 64-byte line
#define PX_DEFAULT_SPINS (1u << 22)               // polls of one flag before giving up: a few seconds
#define PX_DATA_BYTES ((size_t)PX_SLOTS * PX_MAX_WORLD * PX_MAXN * sizeof(double))
#define PX_FLAG_BYTES ((size_t)PX_SLOTS * PX_MAX_WORLD * PX_FLAG_STRIDE * sizeof(unsigned))

struct PxArgs {
    double* vals;                       // in: this rank's n values; out: the sums over the ranks
    char* region[PX_MAX_WORLD];         // every rank's exchange region as mapped into THIS process (region[rank] = own)
    int rank, world, n;
    unsigned seq, spin_limit;
    int* status;                        // device word: 0 ok, 1 = a peer's flag did not arrive in time
};

__global__ __launch_bounds__(256) void peer_allreduce_f64_kernel(const PxArgs a) {
    const int t = threadIdx.x;
    const int slot = (int)(a.seq % PX_SLOTS);
    // ---- push: my values into slot [slot][rank] of every region (my own included: one code path, one sum order) ----
    for (int p = 0; p < a.world; ++p) {
        double* dst = (double*)a.region[p] + ((size_t)slot * PX_MAX_WORLD + a.rank) * PX_MAXN;
        for (int i = t; i < a.n; i += 256) __hip_atomic_store(dst + i, a.vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope: payload before the flags
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < a.world) {
        unsigned* f = (unsigned*)(a.region[t] + PX_DATA_BYTES) + ((size_t)slot * PX_MAX_WORLD + a.rank) * PX_FLAG_STRIDE;
        __hip_atomic_store(f, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---- wait: the flags of all ranks in MY region ----
    __shared__ int timed_out;
    if (t == 0) timed_out = 0;
    __syncthreads();
    if (t < a.world) {
        const unsigned* f = (const unsigned*)(a.region[a.rank] + PX_DATA_BYTES) + ((size_t)slot * PX_MAX_WORLD + t) * PX_FLAG_STRIDE;
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            if (++spins > a.spin_limit) { timed_out = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    if (timed_out) {
        if (t == 0) *a.status = 1;
        return;
    }
    // ---- sum in rank order ----
    const double* src = (const double*)a.region[a.rank] + (size_t)slot * PX_MAX_WORLD * PX_MAXN;
    for (int i = t; i < a.n; i += 256) {
        double s = 0.0;
        for (int r = 0; r < a.world; ++r) s += __hip_atomic_load(src + (size_t)r * PX_MAXN + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.vals[i] = s;
    }
}

extern "C" size_t ic_peer_region_bytes(void) { return PX_DATA_BYTES + PX_FLAG_BYTES; }
extern "C" int ic_peer_max_values(void) { return PX_MAXN; }
extern "C" int ic_peer_max_world(void) { return PX_MAX_WORLD; }

// set-up time (not the hot path): one zeroed fine-grained region on the current device + its inter-process handle (64 bytes)
extern "C" int ic_peer_region_create(void** region, void* ipc_handle_64) {
    IC_CHECK_ARG(region && ipc_handle_64);
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, ic_peer_region_bytes(), hipDeviceMallocFinegrained);
    if (e != hipSuccess) return (int)e;
    if ((e = hipMemset(p, 0, ic_peer_region_bytes())) != hipSuccess) { (void)hipFree(p); return (int)e; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size is part of the ABI");
    if ((e = hipIpcGetMemHandle((hipIpcMemHandle_t*)ipc_handle_64, p)) != hipSuccess) { (void)hipFree(p); return (int)e; }
    *region = p;
    return IC_OK;
}
extern "C" int ic_peer_region_open(const void* ipc_handle_64, void** mapped) {
    IC_CHECK_ARG(ipc_handle_64 && mapped);
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle_64, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(mapped, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? IC_OK : (int)e;
}
extern "C" int ic_peer_region_close(void* mapped) { return mapped ? (int)hipIpcCloseMemHandle(mapped) : IC_ERR_ARG; }
extern "C" int ic_peer_region_destroy(void* region) { return region ? (int)hipFree(region) : IC_ERR_ARG; }

// vals (device, n <= ic_peer_max_values() doubles): in place -> sum over the ranks.  regions_host: `world` pointers, every
// rank's region as mapped in this process (regions_host[rank] = the own one).  seq: 1, 2, 3, ... the same on every rank.
// spin_limit: polls of a peer's flag before the exchange gives up (0 = the default, about a second).
// status: device int (required), set to 1 on a time-out (the values are then left as they were).
extern "C" int ic_peer_allreduce_f64_bounded(double* vals, int n, void* const* regions_host, int rank, int world, unsigned seq,
                                             unsigned spin_limit, int* status, ic_stream_t stream) {
    IC_CHECK_ARG(vals && regions_host && n > 0 && world > 0 && rank >= 0 && rank < world && seq != 0u);
    IC_CHECK_ARG(status != nullptr);                  // the time-out path stores through it
    if (n > PX_MAXN || world > PX_MAX_WORLD) return IC_ERR_UNSUPPORTED;
    PxArgs a{};
    a.vals = vals;
    a.rank = rank;
    a.world = world;
    a.n = n;
    a.seq = seq;
    a.spin_limit = spin_limit ? spin_limit : PX_DEFAULT_SPINS;
    a.status = status;
    for (int r = 0; r < world; ++r) {
        IC_CHECK_ARG(regions_host[r] != nullptr);
        a.region[r] = (char*)regions_host[r];
    }
    hipLaunchKernelGGL(peer_allreduce_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    IC_LAUNCH_CHECK();
    return IC_OK;
}
extern "C" int ic_peer_allreduce_f64(double* vals, int n, void* const* regions_host, int rank, int world, unsigned seq,
                                     int* status, ic_stream_t stream) {
    return ic_peer_allreduce_f64_bounded(vals, n, regions_host, rank, world, seq, 0u, status, stream);
}
