// plan.hip -- native issue plans (declared in include/fatezero_hip.h, machinery in fz_rt.h): the launch list of a UNet forward recorded once
// and re-issued from native code, with the step's pointers patched in.  Replaces, for the steady-state steps of a job, the Python walk over
// the module tree that the reference's `unet(latents, t, encoder_hidden_states=...)` (p2p_ddim_spatial_temporal.py:286, 360-372) is.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

struct FzPlan {
    fz_plan::Plan plan;
    bool paused = false;
#ifndef FZ_EMU
    // the whole plan as ONE executable hipGraph (a chain of kernel nodes in record order): fz_plan_graph_launch
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;
#endif
    std::vector<unsigned char> dirty;   // records whose arguments fz_plan_relocate changed since the graph last saw them
    bool any_dirty = false;
};

#ifndef FZ_EMU
static void fz_node_params(FzPlan* p, size_t i, hipKernelNodeParams* kp, void** ptrs) {
    const fz_plan::Record& r = p->plan.recs[i];
    r.argptrs((unsigned char*)p->plan.args.data() + r.arg_off, ptrs);
    memset(kp, 0, sizeof(*kp));
    kp->func = (void*)r.kernel;
    kp->gridDim = r.grid;
    kp->blockDim = r.block;
    kp->sharedMemBytes = (unsigned)r.smem;
    kp->kernelParams = ptrs;
    kp->extra = nullptr;
}
#endif

extern "C" int fz_plan_begin(FzPlan** out) {
    if (out == nullptr || fz_plan::g_recording != nullptr) return FZ_ERR_BAD_ARG;  // one recording at a time per thread
    FzPlan* p = new FzPlan();
    fz_plan::g_recording = &p->plan;
    *out = p;
    return FZ_OK;
}

// 0: launches are issued without being recorded until fz_plan_pause(p, 0) -- what the host does live at every replay (the controller's
// own launches) stays out of the plan
extern "C" int fz_plan_pause(FzPlan* p, int paused) {
    if (p == nullptr) return FZ_ERR_BAD_ARG;
    if (paused) {
        if (fz_plan::g_recording != &p->plan) return FZ_ERR_BAD_ARG;
        fz_plan::g_recording = nullptr;
    } else {
        if (fz_plan::g_recording != nullptr || !p->paused) return FZ_ERR_BAD_ARG;
        fz_plan::g_recording = &p->plan;
    }
    p->paused = paused != 0;
    return FZ_OK;
}

extern "C" int fz_plan_end(FzPlan* p) {
    if (p == nullptr || (fz_plan::g_recording != &p->plan && !p->paused)) return FZ_ERR_BAD_ARG;
    fz_plan::g_recording = nullptr;
    p->paused = false;
    return FZ_OK;
}

extern "C" int64_t fz_plan_launches(const FzPlan* p) { return p == nullptr ? -1 : (int64_t)p->plan.recs.size(); }

extern "C" int64_t fz_plan_relocate(FzPlan* p, int64_t first, int64_t count, const void* old_base, int64_t nbytes, const void* new_base) {
    if (p == nullptr || first < 0 || count < 0 || first + count > (int64_t)p->plan.recs.size() || nbytes <= 0) return -1;
    const uint64_t lo = (uint64_t)old_base, nb = (uint64_t)new_base;
    int64_t patched = 0;
    const bool track = !p->dirty.empty();  // a graph exists: remember which nodes to refresh
    for (int64_t i = first; i < first + count; ++i) {
        const fz_plan::Record& r = p->plan.recs[i];
        uint64_t* w = p->plan.args.data() + r.arg_off / 8;
        uint64_t* const end = p->plan.args.data() + (r.arg_off + r.arg_len + 7) / 8;
        int64_t here = 0;
        for (; w < end; ++w) {
            const uint64_t d = *w - lo;
            if (d < (uint64_t)nbytes) {
                *w = nb + d;
                ++here;
            }
        }
        if (here && track) {
            p->dirty[i] = 1;
            p->any_dirty = true;
        }
        patched += here;
    }
    return patched;
}

// The whole plan as one hipGraph launch: a chain of kernel nodes in record order, instantiated at the first call; records whose arguments
// fz_plan_relocate changed since are refreshed with hipGraphExecKernelNodeSetParams first.  One runtime call per forward instead of one per
// kernel.  (The CPU emulation build of the tests has no graphs: there this is fz_plan_replay over all records.)
extern "C" int fz_plan_graph_launch(FzPlan* p, void* stream) {
    if (p == nullptr || fz_plan::g_recording != nullptr || p->plan.recs.empty()) return FZ_ERR_BAD_ARG;
#ifdef FZ_EMU
    return fz_plan_replay(p, 0, (int64_t)p->plan.recs.size(), stream);
#else
    const size_t n = p->plan.recs.size();
    hipKernelNodeParams kp;
    void* ptrs[32];
    if (p->exec == nullptr) {
        if (hipGraphCreate(&p->graph, 0) != hipSuccess) return FZ_ERR_LAUNCH;
        p->nodes.resize(n);
        for (size_t i = 0; i < n; ++i) {
            fz_node_params(p, i, &kp, ptrs);
            if (hipGraphAddKernelNode(&p->nodes[i], p->graph, i ? &p->nodes[i - 1] : nullptr, i ? 1 : 0, &kp) != hipSuccess) {
                hipGraphDestroy(p->graph);
                p->graph = nullptr;
                p->nodes.clear();
                (void)hipGetLastError();
                return FZ_ERR_LAUNCH;
            }
        }
        if (hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0) != hipSuccess) {
            hipGraphDestroy(p->graph);
            p->graph = nullptr;
            p->exec = nullptr;
            p->nodes.clear();
            (void)hipGetLastError();
            return FZ_ERR_LAUNCH;
        }
        p->dirty.assign(n, 0);
        p->any_dirty = false;
    } else if (p->any_dirty) {
        for (size_t i = 0; i < n; ++i) {
            if (!p->dirty[i]) continue;
            fz_node_params(p, i, &kp, ptrs);
            if (hipGraphExecKernelNodeSetParams(p->exec, p->nodes[i], &kp) != hipSuccess) return FZ_ERR_LAUNCH;
            p->dirty[i] = 0;
        }
        p->any_dirty = false;
    }
    return hipGraphLaunch(p->exec, (hipStream_t)stream) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
#endif
}

extern "C" int fz_plan_replay(const FzPlan* p, int64_t first, int64_t count, void* stream) {
    if (p == nullptr || first < 0 || count < 0 || first + count > (int64_t)p->plan.recs.size()) return FZ_ERR_BAD_ARG;
    if (fz_plan::g_recording != nullptr) return FZ_ERR_BAD_ARG;  // a replay is not recorded into another plan
    const unsigned char* args = (const unsigned char*)p->plan.args.data();
    const fz_plan::Record* r = p->plan.recs.data() + first;
    for (int64_t i = 0; i < count; ++i, ++r) r->run(*r, args, stream);
    return fz_last_launch_status();
}

extern "C" void fz_plan_destroy(FzPlan* p) {
    if (p == nullptr) return;
    if (fz_plan::g_recording == &p->plan) fz_plan::g_recording = nullptr;
#ifndef FZ_EMU
    if (p->exec != nullptr) hipGraphExecDestroy(p->exec);
    if (p->graph != nullptr) hipGraphDestroy(p->graph);
#endif
    delete p;
}
