// Launch plans (C ABI v10): a recorded sequence of C-ABI launches replayed from C.
//
// The fused HAT training step (studiosr_amd/fasttrain.py; the reference's loop is trainer.py:97-109) is ~530 launches whose arguments never change
// after the first step: every pointer is into static buffers.  Building the argument blocks in Python again for every launch cost 12.8 ms of a 17.3 ms
// step (profiles/r04_*).  The host mirror now records the step once -- sr_plan_create() copies every argument block -- and sr_plan_run() enqueues the
// whole sequence from C: per launch one indirect call into the same entry point Python would have called, on the stream table given at run time
// (slot 0 = the caller's current stream; further slots = side streams), plus event record / wait operations for the cross-stream edges.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "sr_host.h"

namespace {
struct Plan {
    std::vector<SrPlanOp> ops;
    std::vector<char> blob;  // every argument block, 16-byte aligned
    std::vector<hipEvent_t> events;  // created by the first run (a plan can be built where no device is visible)
    int n_events = 0;
    int n_streams = 0;
};
typedef int (*Call1)(const void*, void*);
typedef int (*Call2)(const void*, const void*, void*);
typedef int (*CallI)(const void*, int, void*);
}  // namespace

extern "C" void* sr_plan_create(const SrPlanOp* ops, int n, int n_events) {
    if (!ops || n <= 0 || n_events < 0) {
        sr_set_error("sr_plan_create: bad arguments");
        return nullptr;
    }
    Plan* p = new Plan();
    size_t bytes = 0;
    for (int i = 0; i < n; ++i) bytes += ((size_t)(ops[i].arg_bytes > 0 ? ops[i].arg_bytes : 0) + 15) / 16 * 16 + ((size_t)(ops[i].arg2_bytes > 0 ? ops[i].arg2_bytes : 0) + 15) / 16 * 16;
    p->blob.resize(bytes + 16);
    char* cur = p->blob.data();
    cur += (16 - (reinterpret_cast<uintptr_t>(cur) & 15)) & 15;
    p->ops.assign(ops, ops + n);
    for (int i = 0; i < n; ++i) {
        SrPlanOp& o = p->ops[i];
        const bool call = o.kind == SR_PLAN_CALL1 || o.kind == SR_PLAN_CALL2 || o.kind == SR_PLAN_CALLI;
        const bool ev = o.kind == SR_PLAN_EVENT_RECORD || o.kind == SR_PLAN_STREAM_WAIT;
        if (!(call || ev) || o.stream < 0 || (call && (!o.fn || !o.arg || o.arg_bytes <= 0)) || (o.kind == SR_PLAN_CALL2 && (!o.arg2 || o.arg2_bytes <= 0)) ||
            (ev && (o.ival < 0 || o.ival >= n_events))) {
            sr_set_error("sr_plan_create: bad operation %d (kind %d)", i, o.kind);
            delete p;
            return nullptr;
        }
        if (o.stream + 1 > p->n_streams) p->n_streams = o.stream + 1;
        if (call) {
            memcpy(cur, o.arg, (size_t)o.arg_bytes);
            o.arg = cur;
            cur += ((size_t)o.arg_bytes + 15) / 16 * 16;
            if (o.kind == SR_PLAN_CALL2) {
                memcpy(cur, o.arg2, (size_t)o.arg2_bytes);
                o.arg2 = cur;
                cur += ((size_t)o.arg2_bytes + 15) / 16 * 16;
            }
        }
    }
    p->n_events = n_events;
    return p;
}

extern "C" int sr_plan_streams(const void* plan) { return plan ? static_cast<const Plan*>(plan)->n_streams : 0; }
extern "C" int sr_plan_ops(const void* plan) { return plan ? (int)static_cast<const Plan*>(plan)->ops.size() : 0; }

extern "C" int sr_plan_run(const void* plan, void* const* streams, int n_streams) {
    SR_REQUIRE(plan && streams, "sr_plan_run: null pointer");
    Plan& p = *static_cast<Plan*>(const_cast<void*>(plan));
    SR_REQUIRE(n_streams >= p.n_streams, "sr_plan_run: the plan uses %d streams, %d given", p.n_streams, n_streams);
    while ((int)p.events.size() < p.n_events) {
        hipEvent_t e;
        SR_REQUIRE(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "sr_plan_run: hipEventCreateWithFlags failed");
        p.events.push_back(e);
    }
    for (size_t i = 0; i < p.ops.size(); ++i) {
        const SrPlanOp& o = p.ops[i];
        void* st = streams[o.stream];
        int rc = SR_OK;
        switch (o.kind) {
            case SR_PLAN_CALL1: rc = reinterpret_cast<Call1>(const_cast<void*>(o.fn))(o.arg, st); break;
            case SR_PLAN_CALL2: rc = reinterpret_cast<Call2>(const_cast<void*>(o.fn))(o.arg, o.arg2, st); break;
            case SR_PLAN_CALLI: rc = reinterpret_cast<CallI>(const_cast<void*>(o.fn))(o.arg, o.ival, st); break;
            case SR_PLAN_EVENT_RECORD:
                if (hipEventRecord(p.events[o.ival], reinterpret_cast<hipStream_t>(st)) != hipSuccess) {
                    sr_set_error("sr_plan_run: hipEventRecord failed (operation %zu)", i);
                    rc = SR_ELAUNCH;
                }
                break;
            case SR_PLAN_STREAM_WAIT:
                if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(st), p.events[o.ival], 0) != hipSuccess) {
                    sr_set_error("sr_plan_run: hipStreamWaitEvent failed (operation %zu)", i);
                    rc = SR_ELAUNCH;
                }
                break;
            default: rc = SR_EINVAL;
        }
        if (rc != SR_OK) return rc;  // (the failing entry point has set the error text)
    }
    return SR_OK;
}

// argument-block forms of the positional launches of a training step (recordable as SR_PLAN_CALL1)
extern "C" int sr_tr_add_args(const SrTrAdd* a, void* stream) {
    SR_REQUIRE(a, "sr_tr_add_args: null pointer");
    return sr_tr_add(a->a, a->b, a->b_dtype, a->out, a->n, stream);
}
extern "C" int sr_tr_finalize_to_args(const SrTrFinalize* a, void* stream) {
    SR_REQUIRE(a, "sr_tr_finalize_to_args: null pointer");
    SR_REQUIRE(a->lanes == 0 || a->lanes == 1 || a->lanes == 8, "sr_tr_finalize_to_args: lanes must be 0, 1 or 8");
    if (a->lanes == 8) return sr_tr_finalize_to8(a->arena, a->src, a->dst, a->stride, a->ns, a->scale, a->grad, a->n, stream);
    return sr_tr_finalize_to(a->arena, a->src, a->dst, a->stride, a->ns, a->scale, a->grad, a->n, stream);
}
extern "C" int sr_tr_unshuffle_args(const SrTrUnshuffle* a, void* stream) {
    SR_REQUIRE(a, "sr_tr_unshuffle_args: null pointer");
    return sr_tr_unshuffle(a->src, a->dst, a->B, a->H, a->W, a->cps, a->r, stream);
}
extern "C" int sr_tr_lrelu_bwd_args(const SrTrLreluBwd* a, void* stream) {
    SR_REQUIRE(a, "sr_tr_lrelu_bwd_args: null pointer");
    return sr_tr_lrelu_bwd(a->dy, a->y, a->dx, a->slope, a->n, stream);
}
extern "C" int sr_layernorm_to_args(const SrLayernorm* a, void* stream) {
    SR_REQUIRE(a, "sr_layernorm_to_args: null pointer");
    return sr_layernorm_to(a->x, a->y, a->y_dtype, a->gamma, a->beta, a->M, a->C, a->Cp, a->eps, stream);
}

extern "C" void sr_plan_destroy(void* plan) {
    if (!plan) return;
    Plan* p = static_cast<Plan*>(plan);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    delete p;
}
