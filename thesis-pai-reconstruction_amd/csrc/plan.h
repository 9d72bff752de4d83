// Launch plans: every kernel launch of libpai_hip.so goes through pai::launch.  While a plan is being recorded
// (pai_plan_begin ... pai_plan_end) each launch is executed AND appended to the plan with its kernel handle, grid, block,
// LDS size, stream and a by-value copy of its arguments; pai_plan_run then re-issues the whole sequence -- the ~205
// launches of a Pix2Pix GAN step on their three streams, with the cross-stream event edges recorded through
// pai_stream_wait -- from ONE C call: no descriptor checks, no tap tables, no cost models, no Python.
// (Reference: the step is one Python call, models/wrapper.py:117-162; here it is one C call per recorded step.)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>
#include <tuple>
#include <utility>

namespace pai {

struct PlanOp {
    virtual ~PlanOp() {}
    virtual hipError_t run(int64_t step_delta) = 0;
    virtual int kind() const = 0;      // 0 kernel, 1 memset, 2 event record, 3 stream wait, 4 host call (collective)
    virtual hipStream_t stream() const = 0;
    // kernels only: the launch signals `ev` when it completes (hipExtLaunchKernel's stop event) -- the edge to another
    // stream then needs no marker packet behind it on this stream (pai_stream_wait_last)
    virtual bool set_stop_event(hipEvent_t) { return false; }
    virtual hipEvent_t stop_event() const { return nullptr; }
};

// Adam launches carry two kernel arguments derived from the optimizer step count on the host (lr / (1 - beta1^t),
// 1 / sqrt(1 - beta2^t)): a recorded launch keeps (lr, beta1, beta2, t0) and recomputes both for t0 + step_delta.
struct AdamPatch {
    int active;
    int arg_lr_over_bc1, arg_inv_sqrt_bc2;      // indices into the kernel's argument list
    double lr, beta1, beta2;
    int64_t step0;
};

struct Plan;
extern std::atomic<Plan*> g_recording;      // process-wide: autograd runs the backward pass on a thread of its own
inline bool recording() { return g_recording.load(std::memory_order_acquire) != nullptr; }
void plan_push(PlanOp* op);                 // appends to the plan being recorded (takes ownership); no-op when none is
// The NEXT launch of this thread is an Adam launch whose arguments `a_lr`, `a_bc2` depend on the step count.
void plan_mark_adam(int a_lr, int a_bc2, double lr, double beta1, double beta2, int64_t step);
AdamPatch plan_take_adam();
void adam_coeffs(double lr, double beta1, double beta2, int64_t step, float* lr_over_bc1, float* inv_sqrt_bc2);

template <class... P>
struct KernelOp final : PlanOp {
    void (*k)(P...);
    dim3 g, b;
    unsigned sh;
    hipStream_t st;
    std::tuple<P...> args;
    void* ptrs[sizeof...(P) + 1];
    AdamPatch patch;
    hipEvent_t stop = nullptr;

    template <size_t... I> void bind(std::index_sequence<I...>) { ((ptrs[I] = (void*)&std::get<I>(args)), ...); }
    KernelOp(void (*k_)(P...), dim3 g_, dim3 b_, unsigned sh_, hipStream_t st_, const std::tuple<P...>& a, const AdamPatch& p)
        : k(k_), g(g_), b(b_), sh(sh_), st(st_), args(a), patch(p) {
        bind(std::index_sequence_for<P...>{});
    }
    hipError_t run(int64_t step_delta) override {
        if (patch.active) {
            adam_coeffs(patch.lr, patch.beta1, patch.beta2, patch.step0 + step_delta, (float*)ptrs[patch.arg_lr_over_bc1],
                        (float*)ptrs[patch.arg_inv_sqrt_bc2]);
        }
        if (stop) return hipExtLaunchKernel((const void*)k, g, b, ptrs, sh, st, nullptr, stop, 0);
        return hipLaunchKernel((const void*)k, g, b, ptrs, sh, st);
    }
    int kind() const override { return 0; }
    hipStream_t stream() const override { return st; }
    bool set_stop_event(hipEvent_t ev) override { stop = ev; return true; }
    hipEvent_t stop_event() const override { return stop; }
};

// pai_profile_arm: the next launch of this thread carries the caller's timing events as its OWN start / stop events
// (hipExtLaunchKernel) -- the kernel's duration as the command processor stamps it, which is what rocprofv3 reports, with
// no marker packets in front of and behind the launch (events recorded around a launch read 10-15 % long)
bool profile_take(hipEvent_t* start, hipEvent_t* stop);

template <class... P, size_t... I>
inline hipError_t launch_now(void (*k)(P...), dim3 g, dim3 b, unsigned sh, hipStream_t st, std::tuple<P...>& t,
                             std::index_sequence<I...>) {
    void* ptrs[sizeof...(P) + 1] = {(void*)&std::get<I>(t)...};
    hipEvent_t e0, e1;
    if (profile_take(&e0, &e1)) return hipExtLaunchKernel((const void*)k, g, b, ptrs, sh, st, e0, e1, 0);
    return hipLaunchKernel((const void*)k, g, b, ptrs, sh, st);
}

// kernel<<<grid, block, lds, stream>>>(args...): executed now; appended to the plan being recorded, if any.
template <class... P, class... A>
inline void launch(void (*k)(P...), dim3 g, dim3 b, size_t sh, hipStream_t st, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "pai::launch: argument count does not match the kernel's parameters");
    std::tuple<P...> t(static_cast<P>(a)...);
    if (recording()) {
        AdamPatch patch = plan_take_adam();
        if (patch.active && (patch.arg_lr_over_bc1 >= (int)sizeof...(P) || patch.arg_inv_sqrt_bc2 >= (int)sizeof...(P))) patch.active = 0;
        plan_push(new KernelOp<P...>(k, g, b, (unsigned)sh, st, t, patch));
    }
    (void)launch_now(k, g, b, (unsigned)sh, st, t, std::index_sequence_for<P...>{});   // errors surface in PAI_LAUNCH_CHECK
}

hipError_t memset_async(void* p, int value, size_t bytes, hipStream_t st);

}  // namespace pai

#define PAI_LAUNCH(k, g, b, sh, st, ...) ::pai::launch(k, dim3(g), dim3(b), (size_t)(sh), st, ##__VA_ARGS__)
