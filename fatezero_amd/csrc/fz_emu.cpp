// fz_emu.cpp -- CPU emulation of the HIP execution model (TEST INFRASTRUCTURE ONLY, see fz_rt.h).
//
// One OS thread runs one workgroup at a time; every lane of the workgroup is a fiber with its own
// stack, scheduled round-robin and switched only at collectives (__syncthreads, wave exchanges).
// Deterministic by construction; a missing barrier shows up as a wrong result, not as a flake.
#ifdef FZ_EMU
#include "fz_rt.h"

#include <sys/mman.h>

#include <atomic>
#include <cstdio>
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

struct Fiber {
    void* sp = nullptr;
    bool done = false;
    dim3 tid;
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
    const std::function<void()>* body = nullptr;
    std::vector<unsigned char> dyn;
    ~Worker();
};
static constexpr size_t kSlot = 256;
static thread_local Worker* t_w = nullptr;
Worker::~Worker() {
    if (stacks) munmap(stacks, kStack * kMaxThreads);
    free(wv_buf);
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

void sync_block() {
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
    memcpy(buf + lane * kSlot, mine, bytes);
    wave_barrier(wave, lanes);
    for (int i = 0; i < 64; ++i) memcpy((unsigned char*)all + i * bytes, buf + (i < lanes ? i : 0) * kSlot, bytes);
    wave_barrier(wave, lanes);
}

static Worker* get_worker() {
    static thread_local Worker w;
    if (!w.stacks) {
        w.stacks = (unsigned char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                                        MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        w.wv_buf = (unsigned char*)malloc((kMaxThreads / 64) * 64 * kSlot);
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

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const long total = (long)grid.x * grid.y * grid.z;
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)(hw ? hw : 4);
    const char* env = getenv("FZ_EMU_THREADS");
    if (env) nthreads = atoi(env);
    if (nthreads > total) nthreads = (int)total;
    if (nthreads < 1) nthreads = 1;
    std::atomic<long> next(0);
    auto work = [&]() {
        Worker* w = get_worker();
        for (;;) {
            const long b = next.fetch_add(1);
            if (b >= total) break;
            dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
            run_block(w, grid, block, bid, smem, body);
        }
    };
    if (nthreads == 1) {
        work();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nthreads; ++i) ts.emplace_back(work);
        for (auto& t : ts) t.join();
    }
}

}  // namespace fz_emu
#endif  // FZ_EMU
