// peer.hip -- one-sided exchanges between the GPUs that share ONE frame-sharded clip.
//
// The reference has no multi-GPU path; here the frames of a clip are split over the GPUs of a node and are coupled in four places
// (5-D GroupNorm statistics, resnet.py:338,369; the k=3 temporal convolutions, lora.py:31-54; the sparse-causal K / V^T of neighbour
// and anchor frames, attention.py:374-388; temporal attention over all frames, attention.py:327-337).  Every coupling is a small
// message on the critical path of a sequential layer chain -- ~100 of them per UNet forward -- so what matters is latency, not
// bandwidth: a collective call per message (host enqueue, RCCL kernel launch, proxy hand-shake: tens of microseconds each) would cost
// more than the whole forward on eight GPUs.  Instead every rank owns a SYMMETRIC HEAP that its peers map into their address space
// (hipIpc over xGMI; fatezero_amd/dist.py: PeerHeap) and the exchange is two tiny kernels on the producing / consuming streams:
//
//   fz_peer_put   copies a contiguous message into the same heap offset of up to 8 peers (plain 16-byte stores over xGMI), makes the
//                 data visible (system-scope release) and then publishes the exchange's epoch in the flag word the receivers reserve
//                 for this sender;
//   fz_peer_wait  ONE workgroup that polls the flag words of the senders it expects (system-scope acquire loads, s_sleep between
//                 polls) until each holds an epoch >= the awaited one; the kernels queued behind it on the stream then read the heap.
//
// No host round trip, no collective library on the data path, and the put of a large message (K / V^T panels) runs on a side stream
// under the projections that follow.  The wait is a kernel of its own -- one workgroup, the rest of the chip stays free -- rather than a
// spin inside the consumer: the CPU-less test box runs two ranks on ONE GPU, where a consumer grid that fills the chip while spinning
// would keep the peer's put from ever being scheduled.  Epochs only grow and every (sender, receiver) pair has its own flag word, so
// flags are never reset; a bounded spin (timeout) turns a lost peer into an error code in `err` instead of a hung GPU.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

#define FZ_PEER_MAX 8

struct PeerPutArgs {
    const char* src;
    char* dst[FZ_PEER_MAX];
    uint32_t* flag[FZ_PEER_MAX];
    int n_dst;
    int64_t bytes;
    uint32_t epoch;
    uint32_t* done;  // local counter of workgroups that finished their stores (reset to 0 by the last one)
};

#ifdef FZ_EMU
#include <sched.h>
#include <time.h>
static inline void peer_fence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void peer_store_release(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
static inline uint32_t peer_load_acquire(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
static inline uint32_t peer_add_relaxed(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline void peer_sleep() {
    struct timespec ts = {0, 20000};
    nanosleep(&ts, nullptr);
}
static inline long long peer_clock_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000;
}
#else
FZ_DEVICE void peer_fence_system() { __threadfence_system(); }
FZ_DEVICE void peer_store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
FZ_DEVICE uint32_t peer_load_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
FZ_DEVICE uint32_t peer_add_relaxed(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }
FZ_DEVICE void peer_sleep() { __builtin_amdgcn_s_sleep(64); }
FZ_DEVICE long long peer_clock_us() { return (long long)(wall_clock64() / 100); }  // the constant 100 MHz counter
#endif

FZ_KERNEL void __launch_bounds__(256) peer_put_kernel(PeerPutArgs a) {
    const int64_t chunks = a.bytes >> 4;
    const char* src = a.src;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * 256) {
        const half8_t v = *reinterpret_cast<const half8_t*>(src + (i << 4));
#pragma unroll
        for (int p = 0; p < FZ_PEER_MAX; ++p)
            if (p < a.n_dst) *reinterpret_cast<half8_t*>(a.dst[p] + (i << 4)) = v;
    }
    // every thread's stores are visible system-wide before the workgroup counts itself done; the LAST workgroup publishes the epoch
    peer_fence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = peer_add_relaxed(a.done, 1u);
        if (prev + 1 == gridDim.x) {
            peer_store_release(a.done, 0u);
            peer_fence_system();
            for (int p = 0; p < a.n_dst; ++p) peer_store_release(a.flag[p], a.epoch);
        }
    }
}

// flags: this rank's flag words, one per sender rank; bit r of `mask` = wait for sender r.  err: [0] = 1 when the spin timed out.
FZ_KERNEL void __launch_bounds__(64) peer_wait_kernel(const uint32_t* flags, uint64_t mask, uint32_t epoch, uint32_t* err, int64_t timeout_us) {
    const int r = threadIdx.x;
    if (r < 64 && ((mask >> r) & 1ull)) {
        const long long t0 = peer_clock_us();
        int polls = 0;
        while ((int32_t)(peer_load_acquire(flags + r) - epoch) < 0) {
            peer_sleep();
            if ((++polls & 63) == 0 && timeout_us > 0 && peer_clock_us() - t0 > timeout_us) {
                peer_store_release(err, 1u);
                break;
            }
        }
    }
    peer_fence_system();
}

extern "C" int fz_peer_put(const void* src, int64_t bytes, void* const* dst, uint32_t* const* flags, int n_dst, uint32_t epoch,
                           uint32_t* done_counter, void* stream) {
    if (!src || !dst || !flags || !done_counter || n_dst < 1 || n_dst > FZ_PEER_MAX || bytes <= 0 || (bytes & 15)) return FZ_ERR_BAD_ARG;
    PeerPutArgs a = {};
    a.src = (const char*)src;
    for (int p = 0; p < n_dst; ++p) {
        if (!dst[p] || !flags[p] || ((uintptr_t)dst[p] & 15)) return FZ_ERR_BAD_ARG;
        a.dst[p] = (char*)dst[p];
        a.flag[p] = flags[p];
    }
    if ((uintptr_t)src & 15) return FZ_ERR_BAD_ARG;
    a.n_dst = n_dst;
    a.bytes = bytes;
    a.epoch = epoch;
    a.done = done_counter;
    // small messages (GroupNorm partials: KBs) are ONE workgroup -- latency; panels of MBs get enough workgroups to fill a few xGMI links
    const int64_t chunks = bytes >> 4;
    int blocks = (int)((chunks + 256 * 8 - 1) / (256 * 8));
    blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
    FZ_LAUNCH(peer_put_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return fz_last_launch_status();
}

extern "C" int fz_peer_wait(const uint32_t* flags, uint64_t sender_mask, uint32_t epoch, uint32_t* err, int64_t timeout_us, void* stream) {
    if (!flags || !err) return FZ_ERR_BAD_ARG;
    if (sender_mask == 0) return FZ_OK;
    FZ_LAUNCH(peer_wait_kernel, dim3(1), dim3(64), 0, stream, flags, sender_mask, epoch, err, timeout_us);
    return fz_last_launch_status();
}
