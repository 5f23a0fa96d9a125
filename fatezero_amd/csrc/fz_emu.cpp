// fz_emu.cpp -- CPU emulation of the HIP execution model (TEST INFRASTRUCTURE ONLY, see fz_rt.h).
//
// A persistent pool of OS threads, each running one workgroup at a time; every lane of the workgroup is a fiber with its own
// stack, scheduled round-robin and switched only at collectives (__syncthreads, wave exchanges).
// Deterministic by construction; a missing barrier shows up as a wrong result, not as a flake.
#ifdef FZ_EMU
#include "fz_rt.h"

#include <sys/mman.h>

#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

namespace fz_emu {

thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local unsigned char* t_dyn_smem = nullptr;

extern "C" void fz_emu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl fz_emu_switch
.type fz_emu_switch,@function
fz_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size fz_emu_switch,.-fz_emu_switch
)");

static constexpr size_t kStack = 256 * 1024;
static constexpr int kMaxThreads = 1024;

struct DmaReq {
    const void* src;
    void* dst;
};
static constexpr int kDmaDepth = 64;  // vmcnt is a 6-bit counter

struct Fiber {
    void* sp = nullptr;
    bool done = false;
    dim3 tid;
    DmaReq dma[kDmaDepth];  // this lane's LDS-DMA requests in flight, oldest first from dma_head
    int dma_head = 0, dma_count = 0;
};

struct Worker {
    unsigned char* stacks = nullptr;
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = -1;
    int n = 0, n_live = 0;
    // block barrier
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    // per-wave exchange state
    int wv_arrived[kMaxThreads / 64];
    unsigned wv_gen[kMaxThreads / 64];
    unsigned char* wv_buf = nullptr;  // [waves][64][kSlot]
    float* mm_buf = nullptr;          // [waves][2 (ping-pong)][kMmFloats]: MFMA operands as fp32 + the transposed A panels
    const std::function<void()>* body = nullptr;
    std::vector<unsigned char> dyn;
    ~Worker();
};
static constexpr size_t kSlot = 256;
static constexpr size_t kMmFloats = 64 * 8 * 2 + 2 * 16 * 16;  // A, B fragments of 64 lanes; A^T[k][16 rows] per lane half
static thread_local Worker* t_w = nullptr;
Worker::~Worker() {
    if (stacks) munmap(stacks, kStack * kMaxThreads);
    free(wv_buf);
    free(mm_buf);
}

static void yield_to_sched() {
    Worker* w = t_w;
    Fiber& f = w->fibers[w->cur];
    fz_emu_switch(&f.sp, w->sched_sp);
    t_threadIdx = w->fibers[w->cur].tid;  // restored after being resumed
}

static void fiber_main() {
    Worker* w = t_w;
    (*w->body)();
    Fiber& f = w->fibers[w->cur];
    f.done = true;
    w->n_live--;
    fz_emu_switch(&f.sp, w->sched_sp);
    __builtin_trap();
}

int lane_id() {
    const dim3& b = t_blockDim;
    const dim3& t = t_threadIdx;
    return (int)((t.z * b.y + t.y) * b.x + t.x) & 63;
}
static int linear_tid() {
    const dim3& b = t_blockDim;
    const dim3& t = t_threadIdx;
    return (int)((t.z * b.y + t.y) * b.x + t.x);
}

static bool g_dma_early = false;  // FZ_EMU_DMA=early: requests land at issue (read per launch)

void dma_issue(const void* src16, void* dst16) {
    Fiber& f = t_w->fibers[t_w->cur];
    if (g_dma_early) {
        memcpy(dst16, src16, 16);
        return;
    }
    if (f.dma_count == kDmaDepth) __builtin_trap();  // more than 63 requests outstanding: vmcnt would have wrapped
    f.dma[(f.dma_head + f.dma_count) % kDmaDepth] = {src16, dst16};
    f.dma_count++;
}

void dma_wait(int max_outstanding) {
    Fiber& f = t_w->fibers[t_w->cur];
    while (f.dma_count > max_outstanding) {  // requests retire in order
        const DmaReq& r = f.dma[f.dma_head];
        memcpy(r.dst, r.src, 16);
        f.dma_head = (f.dma_head + 1) % kDmaDepth;
        f.dma_count--;
    }
}

void sync_block() {
    dma_wait(0);
    sync_block_nodrain();
}

void sync_block_nodrain() {
    Worker* w = t_w;
    const unsigned gen = w->bar_gen;
    w->bar_arrived++;
    for (;;) {
        if (w->bar_gen != gen) return;
        if (w->bar_arrived >= w->n_live) {
            w->bar_arrived = 0;
            w->bar_gen++;
            return;
        }
        yield_to_sched();
    }
}

static void wave_barrier(int wave, int lanes_in_wave) {
    Worker* w = t_w;
    const unsigned gen = w->wv_gen[wave];
    w->wv_arrived[wave]++;
    for (;;) {
        if (w->wv_gen[wave] != gen) return;
        if (w->wv_arrived[wave] >= lanes_in_wave) {
            w->wv_arrived[wave] = 0;
            w->wv_gen[wave]++;
            return;
        }
        yield_to_sched();
    }
}

void wave_exchange(const void* mine, void* all, size_t bytes) {
    Worker* w = t_w;
    if (bytes > kSlot) __builtin_trap();
    const int tid = linear_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lanes = (w->n - wave * 64) < 64 ? (w->n - wave * 64) : 64;
    unsigned char* buf = w->wv_buf + (size_t)wave * 64 * kSlot;
    if (bytes == 4) {
        *(uint32_t*)(buf + lane * kSlot) = *(const uint32_t*)mine;
    } else {
        memcpy(buf + lane * kSlot, mine, bytes);
    }
    wave_barrier(wave, lanes);
    if (bytes == 4 && lanes == 64) {  // shuffles / ballots: the common case, without 64 libc calls per lane
        for (int i = 0; i < 64; ++i) ((uint32_t*)all)[i] = *(const uint32_t*)(buf + i * kSlot);
    } else {
        for (int i = 0; i < 64; ++i) memcpy((unsigned char*)all + i * bytes, buf + (i < lanes ? i : 0) * kSlot, bytes);
    }
    wave_barrier(wave, lanes);
}

// v_mfma_f32_32x32x16_f16 for the calling wave.  Fragment semantics (cdna_hip_programming.md section 3):
//   A[i][k]: lane = i + 32*(k/8), element k%8;  B[k][n]: lane = n + 32*(k/8), element k%8
//   C/D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// Every lane converts its own fragments to fp32 once, lanes 0 and 32 lay out the 16 A rows their half of the wave owns as
// A^T[k][reg], and each lane then runs 16 vector FMAs (k ascending: the same summation order as a scalar loop over k).
// The scratch ping-pongs on the wave's barrier generation: two MFMAs with no other collective between them use different
// halves, so a lane that runs ahead never overwrites what a slower lane still reads.
f32x16 mfma_32x32x16_f16(half8_t a, half8_t b, f32x16 c) {
    Worker* w = t_w;
    const int tid = linear_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lanes = (w->n - wave * 64) < 64 ? (w->n - wave * 64) : 64;
    if (lanes != 64) __builtin_trap();  // MFMA kernels launch whole waves
    float* base = w->mm_buf + ((size_t)wave * 2 + ((w->wv_gen[wave] >> 1) & 1)) * kMmFloats;
    float* A = base;
    float* B = base + 64 * 8;
    float* AT = base + 64 * 16;
    for (int e = 0; e < 8; ++e) {
        A[lane * 8 + e] = (float)a[e];
        B[lane * 8 + e] = (float)b[e];
    }
    wave_barrier(wave, 64);
    if ((lane & 31) == 0) {
        float* at = AT + (lane >> 5) * 256;
        for (int k = 0; k < 16; ++k)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                at[k * 16 + r] = A[(row + 32 * (k >> 3)) * 8 + (k & 7)];
            }
    }
    wave_barrier(wave, 64);
    const float* at = AT + (lane >> 5) * 256;
    const int col = lane & 31;
    for (int k = 0; k < 16; ++k) {
        const float bv = B[(col + 32 * (k >> 3)) * 8 + (k & 7)];
        const f32x16 av = *reinterpret_cast<const f32x16*>(at + k * 16);
        c = __builtin_elementwise_fma(av, (f32x16)(bv), c);
    }
    return c;
}

// v_mfma_f32_16x16x32_f16 for the calling wave:  A[i][k]: lane = i + 16*(k/8), element k%8;  B[k][n]: lane = n + 16*(k/8), element k%8;
// C/D[row][col]: col = lane&15, row = 4*(lane>>4) + reg.  k ascending per output element.  Two wave barriers per call, like the
// 32x32x16 routine: the scratch half alternates per MFMA with the barrier generation.
f32x4 mfma_16x16x32_f16(half8_t a, half8_t b, f32x4 c) {
    Worker* w = t_w;
    const int tid = linear_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lanes = (w->n - wave * 64) < 64 ? (w->n - wave * 64) : 64;
    if (lanes != 64) __builtin_trap();
    float* base = w->mm_buf + ((size_t)wave * 2 + ((w->wv_gen[wave] >> 1) & 1)) * kMmFloats;
    float* A = base;
    float* B = base + 64 * 8;
    for (int e = 0; e < 8; ++e) {
        A[lane * 8 + e] = (float)a[e];
        B[lane * 8 + e] = (float)b[e];
    }
    wave_barrier(wave, 64);
    wave_barrier(wave, 64);
    const int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k)
            acc = __builtin_fmaf(A[(row + 16 * (k >> 3)) * 8 + (k & 7)], B[(col + 16 * (k >> 3)) * 8 + (k & 7)], acc);
        c[r] = acc;
    }
    return c;
}

static Worker* get_worker() {
    static thread_local Worker w;
    if (!w.stacks) {
        w.stacks = (unsigned char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                        MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        w.wv_buf = (unsigned char*)malloc((kMaxThreads / 64) * 64 * kSlot);
        w.mm_buf = (float*)aligned_alloc(64, (kMaxThreads / 64) * 2 * kMmFloats * sizeof(float));
        w.fibers.resize(kMaxThreads);
    }
    return &w;
}

static void run_block(Worker* w, dim3 grid, dim3 block, dim3 bid, size_t smem, const std::function<void()>& body) {
    t_w = w;
    const int n = (int)(block.x * block.y * block.z);
    if (n > kMaxThreads) __builtin_trap();
    w->n = w->n_live = n;
    w->bar_arrived = 0;
    w->body = &body;
    memset(w->wv_arrived, 0, sizeof(w->wv_arrived));
    w->dyn.assign(smem + 64, 0);
    t_dyn_smem = (unsigned char*)(((uintptr_t)w->dyn.data() + 15) & ~(uintptr_t)15);
    t_gridDim = grid;
    t_blockDim = block;
    t_blockIdx = bid;
    for (int i = 0; i < n; ++i) {
        Fiber& f = w->fibers[i];
        f.done = false;
        f.dma_head = f.dma_count = 0;
        f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
        // initial stack: 6 callee-saved slots + return address (fiber_main); keep 16-B alignment at entry
        uintptr_t top = (uintptr_t)(w->stacks + (size_t)(i + 1) * kStack);
        top &= ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;              // fake return address of fiber_main (never used)
        *--sp = (void*)&fiber_main;   // `ret` target of the first switch
        for (int k = 0; k < 6; ++k) *--sp = nullptr;
        f.sp = sp;
    }
    while (w->n_live > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = w->fibers[i];
            if (f.done) continue;
            w->cur = i;
            t_threadIdx = f.tid;
            fz_emu_switch(&w->sched_sp, f.sp);
        }
    }
}

// Persistent pool of OS threads: each keeps its Worker (fiber stacks mapped and warm) across launches -- spawning the threads per
// launch cost ~20 ms (mmap + first-touch faults of every fiber stack) on kernels whose emulated work is microseconds.
struct Pool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t job_id = 0;
    int pending = 0, size = 0;
    pid_t pid = 0;
    // the job being run
    dim3 grid, block;
    size_t smem = 0;
    const std::function<void()>* body = nullptr;
    long total = 0;
    int limit = 0;  // pool threads with index < limit take blocks
    std::atomic<long> next{0};
};
static Pool* g_pool = nullptr;     // leaked on purpose: its threads are detached and die with the process
static std::mutex g_launch_mu;     // one launch at a time

static void drain(Pool* p) {
    Worker* w = get_worker();
    for (;;) {
        const long b = p->next.fetch_add(1);
        if (b >= p->total) break;
        const dim3& g = p->grid;
        dim3 bid((unsigned)(b % g.x), (unsigned)((b / g.x) % g.y), (unsigned)(b / ((long)g.x * g.y)));
        run_block(w, g, p->block, bid, p->smem, *p->body);
    }
}

static void pool_thread(Pool* p, int index) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->job_id != seen; });
            seen = p->job_id;
        }
        if (index < p->limit) drain(p);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (--p->pending == 0) p->cv_done.notify_one();
        }
    }
}

static Pool* get_pool(int nthreads) {
    if (g_pool && g_pool->pid == getpid() && g_pool->size >= nthreads) return g_pool;
    // first launch, a forked child (the parent's threads do not exist here), or a larger FZ_EMU_THREADS: a fresh pool
    Pool* p = new Pool();
    p->pid = getpid();
    p->size = nthreads;
    for (int i = 0; i < nthreads; ++i) std::thread(pool_thread, p, i).detach();
    g_pool = p;
    return p;
}

static std::atomic<int> g_null_launch{0};   // host-issue-time measurements: every launch returns at once (fz_emu_set_null_launch)
static std::atomic<long> g_launch_count{0};

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    g_launch_count.fetch_add(1);
    if (g_null_launch.load()) return;
    const long total = (long)grid.x * grid.y * grid.z;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)(hw ? hw : 4);
    const char* env = getenv("FZ_EMU_THREADS");
    if (env) nthreads = atoi(env);
    if (nthreads < 1) nthreads = 1;
    std::lock_guard<std::mutex> launch_lk(g_launch_mu);
    const char* dma_mode = getenv("FZ_EMU_DMA");
    g_dma_early = dma_mode && dma_mode[0] == 'e';
    if (nthreads == 1 || total <= 1) {  // the calling thread alone
        Pool one;
        one.grid = grid; one.block = block; one.smem = smem; one.body = &body; one.total = total;
        drain(&one);
        return;
    }
    Pool* p = get_pool(nthreads - 1);  // the calling thread works too
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->grid = grid; p->block = block; p->smem = smem; p->body = &body; p->total = total;
        p->limit = (int)(total - 1 < nthreads - 1 ? total - 1 : nthreads - 1);
        p->next.store(0);
        p->pending = p->size;  // every pool thread acknowledges the job, so the next launch cannot overtake a sleeper
        p->job_id++;
    }
    p->cv_job.notify_all();
    drain(p);
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->pending == 0; });
}

}  // namespace fz_emu

// Emulator-only switches (not part of include/fatezero_hip.h): with null launches on, every fz_* call does its host work and returns without
// running the kernel -- what is left is the host's issue time; the counter counts kernel launches either way.
extern "C" void fz_emu_set_null_launch(int on) { fz_emu::g_null_launch.store(on); }
extern "C" long fz_emu_launch_count() { return fz_emu::g_launch_count.load(); }
#endif  // FZ_EMU
