// pai_comm_* / pai_allreduce: gradient exchange straight on RCCL (SURVEY 8(b) export set, 8(e): all-reduce of the
// gradient arenas over xGMI, one process per GPU).  librccl is opened lazily with dlopen on the first pai_comm_* call:
// libpai_hip.so has no link-time dependency on it, single-GPU users never load it, and a process that already runs
// torch.distributed (whose librccl.so is in the address space) gets that same library.
#include <dlfcn.h>
#include <stdlib.h>

#include "common.h"

namespace {
struct nccl_id { char internal[128]; };      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
typedef int (*get_id_fn)(nccl_id*);
typedef int (*init_rank_fn)(void**, int, nccl_id, int);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*destroy_fn)(void*);
typedef const char* (*errstr_fn)(int);

struct Rccl {
    void* handle = nullptr;
    get_id_fn get_id = nullptr;
    init_rank_fn init_rank = nullptr;
    allreduce_fn allreduce = nullptr;
    destroy_fn destroy = nullptr;
    errstr_fn errstr = nullptr;
} g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char* names[] = {getenv("PAI_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    PAI_CHECK(h != nullptr, "pai_comm: cannot open librccl.so (%s); set PAI_RCCL_LIB", dlerror());
    g_rccl.get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (init_rank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
    g_rccl.errstr = (errstr_fn)dlsym(h, "ncclGetErrorString");
    PAI_CHECK(g_rccl.get_id && g_rccl.init_rank && g_rccl.allreduce && g_rccl.destroy,
              "pai_comm: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    g_rccl.handle = h;
    return 0;
}

const char* rccl_err(int rc) { return g_rccl.errstr ? g_rccl.errstr(rc) : "?"; }

// a recorded pai_allreduce (launch plans, plan.h): the same in-place collective on the same stream at every replay
struct AllReduceOp final : pai::PlanOp {
    void* comm; void* ptr; size_t count; int nccl_dtype; hipStream_t st;
    AllReduceOp(void* c, void* p, size_t n, int dt, hipStream_t s) : comm(c), ptr(p), count(n), nccl_dtype(dt), st(s) {}
    hipError_t run(int64_t) override {
        return g_rccl.allreduce(ptr, ptr, count, nccl_dtype, 0, comm, st) == 0 ? hipSuccess : hipErrorUnknown;
    }
    int kind() const override { return 4; }
    hipStream_t stream() const override { return st; }
};
}  // namespace

extern "C" int pai_comm_unique_id(void* id_out) {
    PAI_CHECK(id_out != nullptr, "pai_comm_unique_id: null pointer");
    if (rccl_load()) return 1;
    const int rc = g_rccl.get_id((nccl_id*)id_out);
    PAI_CHECK(rc == 0, "ncclGetUniqueId: %s", rccl_err(rc));
    return 0;
}

extern "C" int pai_comm_init(const void* id, int rank, int world, void** comm_out) {
    PAI_CHECK(id && comm_out && world >= 1 && rank >= 0 && rank < world, "pai_comm_init: bad arguments");
    if (rccl_load()) return 1;
    nccl_id uid;
    memcpy(&uid, id, sizeof(uid));
    void* comm = nullptr;
    const int rc = g_rccl.init_rank(&comm, world, uid, rank);
    PAI_CHECK(rc == 0 && comm, "ncclCommInitRank(rank %d of %d): %s", rank, world, rccl_err(rc));
    *comm_out = comm;
    return 0;
}

extern "C" int pai_allreduce(void* comm, void* ptr, int64_t count, int dtype, void* stream) {
    PAI_CHECK(comm && ptr && count >= 0, "pai_allreduce: bad arguments");
    PAI_CHECK(dtype == PAI_F32 || dtype == PAI_BF16, "pai_allreduce: bad dtype %d", dtype);
    PAI_CHECK(g_rccl.handle != nullptr, "pai_allreduce: no communicator was created in this process");
    if (count == 0) return 0;
    // ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0 (rccl.h); in place
    if (pai::recording()) pai::plan_push(new AllReduceOp(comm, ptr, (size_t)count, dtype == PAI_F32 ? 7 : 9, (hipStream_t)stream));
    const int rc = g_rccl.allreduce(ptr, ptr, (size_t)count, dtype == PAI_F32 ? 7 : 9, 0, comm, (hipStream_t)stream);
    PAI_CHECK(rc == 0, "ncclAllReduce: %s", rccl_err(rc));
    return 0;
}

extern "C" int pai_comm_destroy(void* comm) {
    if (!comm) return 0;
    PAI_CHECK(g_rccl.handle != nullptr, "pai_comm_destroy: no communicator was created in this process");
    const int rc = g_rccl.destroy(comm);
    PAI_CHECK(rc == 0, "ncclCommDestroy: %s", rccl_err(rc));
    return 0;
}
