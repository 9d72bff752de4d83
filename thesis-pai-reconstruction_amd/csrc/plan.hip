// pai_plan_* / pai_stream_wait: C-side launch plans (see plan.h).
//
// The reference's training step is ONE Python call that dispatches ~170 ATen operators (models/wrapper.py:117-162); the
// eager step of this library is ~205 kernel launches issued one by one from Python through ctypes (5.5 ms of host time
// under a 6.4 ms step, round 3).  A plan is that launch sequence recorded once per (shape, buffer set): kernel handles,
// launch geometry and argument blocks by value, on the same streams, with the fork / join edges between the streams as
// event record + stream wait pairs.  pai_plan_run walks it in C: one hipLaunchKernel per node, nothing else.
// Unlike a hipGraph replay the launches go to the SAME streams in the SAME order as the eager step, so the co-scheduling
// of the weight-gradient stream with the input-gradient chain -- which a replayed hipGraph serialised (round 2/3:
// 4-6 % slower than eager) -- is unchanged.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace pai {

struct Plan {
    std::vector<PlanOp*> ops;
    std::vector<hipEvent_t> events;      // owned: one per recorded stream wait
    std::mutex mu;
    int device = -1;
    int64_t runs = 0;
    bool sealed = false;
    ~Plan() {
        for (PlanOp* op : ops) delete op;
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
};

std::atomic<Plan*> g_recording{nullptr};

// Events that order one stream of the device behind another need no SYSTEM-scope fence (nothing on the host inspects
// them): with hipEventDisableSystemFence the marker packet releases at agent scope only (PAI_EVENT_FLAGS overrides the
// flag word for A/B timing).
static unsigned event_flags() {
    static const unsigned f = getenv("PAI_EVENT_FLAGS") ? (unsigned)strtoul(getenv("PAI_EVENT_FLAGS"), nullptr, 0)
                                                         : (hipEventDisableTiming | hipEventDisableSystemFence);
    return f;
}
static thread_local AdamPatch t_adam = {0, 0, 0, 0.0, 0.0, 0.0, 0};

void plan_push(PlanOp* op) {
    Plan* p = g_recording.load(std::memory_order_acquire);
    if (!p) {
        delete op;
        return;
    }
    std::lock_guard<std::mutex> lk(p->mu);
    p->ops.push_back(op);
}

static thread_local hipEvent_t t_prof[2] = {nullptr, nullptr};
static std::atomic<int> g_prof_armed{0};      // fast path: no thread has armed anything

bool profile_take(hipEvent_t* start, hipEvent_t* stop) {
    if (g_prof_armed.load(std::memory_order_relaxed) == 0 || !t_prof[1]) return false;
    *start = t_prof[0];
    *stop = t_prof[1];
    t_prof[0] = t_prof[1] = nullptr;
    g_prof_armed.fetch_sub(1, std::memory_order_relaxed);
    return true;
}

static void profile_arm(hipEvent_t start, hipEvent_t stop) {
    if (t_prof[1]) g_prof_armed.fetch_sub(1, std::memory_order_relaxed);      // an unused arming is replaced
    t_prof[0] = start;
    t_prof[1] = stop;
    if (stop) g_prof_armed.fetch_add(1, std::memory_order_relaxed);
}

void plan_mark_adam(int a_lr, int a_bc2, double lr, double beta1, double beta2, int64_t step) {
    // only while a plan is being recorded: the mark is consumed by the launch that follows on this thread, and a mark
    // left behind by an un-recorded call would be applied to whatever kernel this thread records next
    if (recording()) t_adam = {1, a_lr, a_bc2, lr, beta1, beta2, step};
    else t_adam.active = 0;
}

AdamPatch plan_take_adam() {
    AdamPatch p = t_adam;
    t_adam.active = 0;
    return p;
}

// same expressions as pai_adam (misc.hip): the replayed launch of step t is bit-identical to the eager one
void adam_coeffs(double lr, double beta1, double beta2, int64_t step, float* lr_over_bc1, float* inv_sqrt_bc2) {
    const double bc1 = 1.0 - pow(beta1, (int)step);
    const double bc2 = 1.0 - pow(beta2, (int)step);
    *lr_over_bc1 = (float)(lr / bc1);
    *inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

struct MemsetOp final : PlanOp {
    void* p; int value; size_t bytes; hipStream_t st;
    MemsetOp(void* p_, int v, size_t n, hipStream_t s) : p(p_), value(v), bytes(n), st(s) {}
    hipError_t run(int64_t) override { return hipMemsetAsync(p, value, bytes, st); }
    int kind() const override { return 1; }
    hipStream_t stream() const override { return st; }
};

hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t st) {
    if (recording()) plan_push(new MemsetOp(p, value, bytes, st));
    return hipMemsetAsync(p, value, bytes, st);
}

struct RecordOp final : PlanOp {
    hipEvent_t ev; hipStream_t st;
    RecordOp(hipEvent_t e, hipStream_t s) : ev(e), st(s) {}
    hipError_t run(int64_t) override { return hipEventRecord(ev, st); }
    int kind() const override { return 2; }
    hipStream_t stream() const override { return st; }
};

struct WaitOp final : PlanOp {
    hipEvent_t ev; hipStream_t st;
    WaitOp(hipEvent_t e, hipStream_t s) : ev(e), st(s) {}
    hipError_t run(int64_t) override { return hipStreamWaitEvent(st, ev, 0); }
    int kind() const override { return 3; }
    hipStream_t stream() const override { return st; }
};

// eager edges: a ring of events per device, created together on first use (none is created later, e.g. inside a hipGraph
// capture).  A wait captures the record that precedes it, so an event may be re-recorded while an older wait on it is
// still pending.
static const int RING = 256, MAX_DEV = 16;
static hipEvent_t g_ring[MAX_DEV][RING];
static std::atomic<bool> g_ring_made[MAX_DEV];
static std::atomic<unsigned> g_ring_next{0};
static std::mutex g_ring_mu;

static hipError_t ring_event(hipEvent_t* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEV) return hipErrorInvalidDevice;
    if (!g_ring_made[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_ring_mu);
        if (!g_ring_made[dev].load(std::memory_order_relaxed)) {
            for (int i = 0; i < RING; ++i) {
                e = hipEventCreateWithFlags(&g_ring[dev][i], event_flags());
                if (e != hipSuccess) return e;
            }
            g_ring_made[dev].store(true, std::memory_order_release);
        }
    }
    *out = g_ring[dev][g_ring_next.fetch_add(1) % RING];
    return hipSuccess;
}

}  // namespace pai

using pai::Plan;

extern "C" int pai_stream_wait(void* waiting_stream, void* signalling_stream) {
    hipStream_t w = (hipStream_t)waiting_stream, s = (hipStream_t)signalling_stream;
    if (w == s) return 0;
    Plan* p = pai::g_recording.load(std::memory_order_acquire);
    hipEvent_t ev = nullptr;
    hipError_t e;
    if (p) {
        e = hipEventCreateWithFlags(&ev, pai::event_flags());
        PAI_CHECK(e == hipSuccess, "pai_stream_wait: hipEventCreate: %s", hipGetErrorString(e));
        std::lock_guard<std::mutex> lk(p->mu);
        p->events.push_back(ev);
        p->ops.push_back(new pai::RecordOp(ev, s));
        p->ops.push_back(new pai::WaitOp(ev, w));
    } else {
        e = pai::ring_event(&ev);
        PAI_CHECK(e == hipSuccess, "pai_stream_wait: no event: %s", hipGetErrorString(e));
    }
    e = hipEventRecord(ev, s);
    PAI_CHECK(e == hipSuccess, "pai_stream_wait: hipEventRecord: %s", hipGetErrorString(e));
    e = hipStreamWaitEvent(w, ev, 0);
    PAI_CHECK(e == hipSuccess, "pai_stream_wait: hipStreamWaitEvent: %s", hipGetErrorString(e));
    return 0;
}

// pai_stream_wait for an edge whose source is the LAST launch this library made on `signalling_stream` (the caller
// promises that nothing else was enqueued on that stream since).  Executed now it is the same record + wait.  In a plan it
// is recorded differently: the source launch itself carries the event (hipExtLaunchKernel's stop event) and only the wait
// is a node -- no marker packet sits between the source launch and its successor on the signalling stream, which in the
// replayed step cost the main stream a ~5 us bubble in front of every input-gradient launch (one fork per layer).
extern "C" int pai_stream_wait_last(void* waiting_stream, void* signalling_stream) {
    hipStream_t w = (hipStream_t)waiting_stream, s = (hipStream_t)signalling_stream;
    if (w == s) return 0;
    Plan* p = pai::g_recording.load(std::memory_order_acquire);
    static const bool off = getenv("PAI_NO_STOP_EVENTS") && atoi(getenv("PAI_NO_STOP_EVENTS")) != 0;
    if (!p || off) return pai_stream_wait(waiting_stream, signalling_stream);
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        pai::PlanOp* src = nullptr;
        for (size_t i = p->ops.size(); i-- > 0;) {
            if (p->ops[i]->stream() != s) continue;
            if (p->ops[i]->kind() == 0) src = p->ops[i];     // the last node of that stream is a kernel launch
            break;
        }
        if (src) {
            ev = src->stop_event();
            if (!ev) {
                const hipError_t e = hipEventCreateWithFlags(&ev, pai::event_flags());
                PAI_CHECK(e == hipSuccess, "pai_stream_wait_last: hipEventCreate: %s", hipGetErrorString(e));
                p->events.push_back(ev);
                src->set_stop_event(ev);
            }
            p->ops.push_back(new pai::WaitOp(ev, w));
        }
    }
    if (!ev) return pai_stream_wait(waiting_stream, signalling_stream);     // no launch of that stream in this plan yet
    // the step being recorded runs now: an ordinary marker on this event
    hipError_t e = hipEventRecord(ev, s);
    PAI_CHECK(e == hipSuccess, "pai_stream_wait_last: hipEventRecord: %s", hipGetErrorString(e));
    e = hipStreamWaitEvent(w, ev, 0);
    PAI_CHECK(e == hipSuccess, "pai_stream_wait_last: hipStreamWaitEvent: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pai_event_create(pai_event_t* out) {
    PAI_CHECK(out != nullptr, "pai_event_create: null pointer");
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, pai::event_flags());
    PAI_CHECK(e == hipSuccess, "pai_event_create: %s", hipGetErrorString(e));
    *out = (pai_event_t)ev;
    return 0;
}

extern "C" int pai_event_create_timing(pai_event_t* out) {
    PAI_CHECK(out != nullptr, "pai_event_create_timing: null pointer");
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreate(&ev);
    PAI_CHECK(e == hipSuccess, "pai_event_create_timing: %s", hipGetErrorString(e));
    *out = (pai_event_t)ev;
    return 0;
}

extern "C" int pai_event_elapsed_ms(pai_event_t start, pai_event_t stop, float* ms) {
    PAI_CHECK(start && stop && ms, "pai_event_elapsed_ms: null pointer");
    const hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    PAI_CHECK(e == hipSuccess, "pai_event_elapsed_ms: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pai_profile_arm(pai_event_t start, pai_event_t stop) {
    PAI_CHECK((start == nullptr) == (stop == nullptr), "pai_profile_arm: both events or neither");
    pai::profile_arm((hipEvent_t)start, (hipEvent_t)stop);
    return 0;
}

extern "C" int pai_event_destroy(pai_event_t ev) {
    if (!ev) return 0;
    const hipError_t e = hipEventDestroy((hipEvent_t)ev);
    PAI_CHECK(e == hipSuccess, "pai_event_destroy: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pai_event_record(pai_event_t ev, void* stream) {
    PAI_CHECK(ev != nullptr, "pai_event_record: null event");
    if (pai::recording()) pai::plan_push(new pai::RecordOp((hipEvent_t)ev, (hipStream_t)stream));
    const hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
    PAI_CHECK(e == hipSuccess, "pai_event_record: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pai_stream_wait_event(void* waiting_stream, pai_event_t ev) {
    PAI_CHECK(ev != nullptr, "pai_stream_wait_event: null event");
    if (pai::recording()) pai::plan_push(new pai::WaitOp((hipEvent_t)ev, (hipStream_t)waiting_stream));
    const hipError_t e = hipStreamWaitEvent((hipStream_t)waiting_stream, (hipEvent_t)ev, 0);
    PAI_CHECK(e == hipSuccess, "pai_stream_wait_event: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int pai_plan_create(pai_plan_t* out) {
    PAI_CHECK(out != nullptr, "pai_plan_create: null pointer");
    Plan* p = new Plan();
    (void)hipGetDevice(&p->device);
    *out = (pai_plan_t)p;
    return 0;
}

extern "C" int pai_plan_destroy(pai_plan_t plan) {
    Plan* p = (Plan*)plan;
    if (!p) return 0;
    PAI_CHECK(pai::g_recording.load() != p, "pai_plan_destroy: the plan is being recorded");
    delete p;
    return 0;
}

extern "C" int pai_plan_begin(pai_plan_t plan) {
    Plan* p = (Plan*)plan;
    PAI_CHECK(p != nullptr, "pai_plan_begin: null plan");
    PAI_CHECK(!p->sealed, "pai_plan_begin: the plan has already been recorded");
    Plan* none = nullptr;
    PAI_CHECK(pai::g_recording.compare_exchange_strong(none, p), "pai_plan_begin: another plan is being recorded");
    (void)pai::plan_take_adam();
    return 0;
}

extern "C" int pai_plan_end(pai_plan_t plan) {
    Plan* p = (Plan*)plan;
    PAI_CHECK(p != nullptr && pai::g_recording.load() == p, "pai_plan_end: this plan is not being recorded");
    pai::g_recording.store(nullptr, std::memory_order_release);
    std::lock_guard<std::mutex> lk(p->mu);
    p->sealed = true;
    return 0;
}

extern "C" int pai_plan_run(pai_plan_t plan, int64_t step_delta) {
    Plan* p = (Plan*)plan;
    PAI_CHECK(p != nullptr && p->sealed, "pai_plan_run: the plan has not been recorded (pai_plan_begin / pai_plan_end)");
    PAI_CHECK(step_delta >= 0, "pai_plan_run: step_delta %lld < 0", (long long)step_delta);
    // the recorded streams, events and function attributes belong to one device: replaying with another one current would
    // launch on foreign streams
    int dev = -1;
    PAI_CHECK(hipGetDevice(&dev) == hipSuccess && (p->device < 0 || dev == p->device),
              "pai_plan_run: the plan was recorded on device %d, the current device is %d", p->device, dev);
    const size_t n = p->ops.size();
    for (size_t i = 0; i < n; ++i) {
        const hipError_t e = p->ops[i]->run(step_delta);
        if (e != hipSuccess) {
            pai_set_error("pai_plan_run: node %zu of %zu (kind %d) failed: %s", i, n, p->ops[i]->kind(), hipGetErrorString(e));
            return 2;
        }
    }
    ++p->runs;
    return 0;
}

extern "C" int pai_plan_info(pai_plan_t plan, int* launches, int* waits, int* streams, int64_t* runs) {
    Plan* p = (Plan*)plan;
    PAI_CHECK(p != nullptr, "pai_plan_info: null plan");
    std::lock_guard<std::mutex> lk(p->mu);
    int nl = 0, nw = 0;
    std::vector<hipStream_t> seen;
    for (pai::PlanOp* op : p->ops) {
        if (op->kind() <= 1 || op->kind() == 4) ++nl;
        if (op->kind() == 3) ++nw;
        bool found = false;
        for (hipStream_t s : seen) found = found || s == op->stream();
        if (!found) seen.push_back(op->stream());
    }
    if (launches) *launches = nl;
    if (waits) *waits = nw;
    if (streams) *streams = (int)seen.size();
    if (runs) *runs = p->runs;
    return 0;
}
