// plan.hip -- native issue plans (declared in include/fatezero_hip.h, machinery in fz_rt.h): the launch list of a UNet forward recorded once
// and re-issued from native code, with the step's pointers patched in.  Replaces, for the steady-state steps of a job, the Python walk over
// the module tree that the reference's `unet(latents, t, encoder_hidden_states=...)` (p2p_ddim_spatial_temporal.py:286, 360-372) is.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

namespace fz_plan {
Plan* g_recording = nullptr;
}

struct FzPlan {
    fz_plan::Plan plan;
    bool paused = false;
};

extern "C" int fz_plan_begin(FzPlan** out) {
    if (out == nullptr || fz_plan::g_recording != nullptr) return FZ_ERR_BAD_ARG;  // one recording at a time
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
    if (count == 0) return 0;
    // the records of a range own a contiguous run of argument words
    const fz_plan::Record& r0 = p->plan.recs[first];
    const fz_plan::Record& r1 = p->plan.recs[first + count - 1];
    uint64_t* w = p->plan.args.data() + r0.arg_off / 8;
    uint64_t* const end = p->plan.args.data() + (r1.arg_off + r1.arg_len + 7) / 8;
    for (; w < end; ++w) {
        const uint64_t d = *w - lo;
        if (d < (uint64_t)nbytes) {
            *w = nb + d;
            ++patched;
        }
    }
    return patched;
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
    delete p;
}
