// Data-parallel entry points: an RCCL communicator per process and gradient-bucket all-reduces enqueued on the
// caller's stream.  They stand where the reference relies on DistributedDataParallel's reducer
// (/root/reference/setup_training.py:39-41, /root/reference/trainer.py:74-77: one process per device, gradients
// summed across ranks during backward).
//
// RCCL is bound at run time, not at link time: the first call looks for a librccl.so.1 that the process has loaded
// already (PyTorch brings one) and only then loads one itself, so the library has no RCCL dependency for single-GPU
// users and a process never holds two RCCL instances.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

// The slice of the RCCL C API used here, spelled out so that no RCCL header is needed to build
// (rccl.h: ncclUniqueId is 128 opaque bytes passed by value; ncclSum = 0; ncclFloat32 = 7, ncclBfloat16 = 9).
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*CommCountFn)(Comm, int*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    CommCountFn comm_count = nullptr;
    AllReduceFn all_reduce = nullptr;
    ErrStrFn err_str = nullptr;
    bool ok = false;
};

std::mutex g_mu;
Rccl g_rccl;
bool g_tried = false;

const Rccl* rccl() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_tried) return g_rccl.ok ? &g_rccl : nullptr;
    g_tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the instance already in the process
    for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return nullptr;
    g_rccl.handle = h;
    g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    g_rccl.comm_count = (CommCountFn)dlsym(h, "ncclCommCount");
    g_rccl.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    g_rccl.err_str = (ErrStrFn)dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.comm_count &&
                g_rccl.all_reduce && g_rccl.err_str;
    return g_rccl.ok ? &g_rccl : nullptr;
}

int rccl_fail(const Rccl* r, const char* what, int rc) {
    return mi355_fail(MI355_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc, r->err_str(rc));
}

struct DpComm {
    unsigned magic;
    Comm comm;
    int rank, world, device;
};
constexpr unsigned DP_MAGIC = 0x44504331u;

}  // namespace

extern "C" {

int mi355_dp_unique_id_bytes(void) { return (int)sizeof(UniqueId); }

int mi355_dp_unique_id(void* out, int bytes) {
    if (!out || bytes != (int)sizeof(UniqueId)) return mi355_fail(MI355_ERR_ARG, "dp_unique_id: need a %d-byte buffer", (int)sizeof(UniqueId));
    const Rccl* r = rccl();
    if (!r) return mi355_fail(MI355_ERR_UNSUPPORTED, "dp: librccl.so.1 not found or incomplete");
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "dp_unique_id", rc);
    memcpy(out, &id, sizeof(id));
    return MI355_OK;
}

int mi355_dp_comm_create(void** comm, const void* unique_id, int bytes, int rank, int world) {
    if (!comm || !unique_id || bytes != (int)sizeof(UniqueId)) return mi355_fail(MI355_ERR_ARG, "dp_comm_create: bad arguments");
    if (world < 1 || rank < 0 || rank >= world) return mi355_fail(MI355_ERR_ARG, "dp_comm_create: rank %d of %d", rank, world);
    const Rccl* r = rccl();
    if (!r) return mi355_fail(MI355_ERR_UNSUPPORTED, "dp: librccl.so.1 not found or incomplete");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return mi355_fail(MI355_ERR_LAUNCH, "dp_comm_create: no current device");
    UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    Comm c = nullptr;
    const int rc = r->comm_init_rank(&c, world, id, rank);
    if (rc != 0) return rccl_fail(r, "dp_comm_create", rc);
    int n = 0;
    const int rc2 = r->comm_count(c, &n);
    if (rc2 != 0 || n != world) {
        (void)r->comm_destroy(c);
        return mi355_fail(MI355_ERR_LAUNCH, "dp_comm_create: communicator reports %d ranks, expected %d", n, world);
    }
    DpComm* d = new DpComm{DP_MAGIC, c, rank, world, dev};
    *comm = d;
    return MI355_OK;
}

int mi355_dp_comm_world(void* comm) {
    DpComm* d = (DpComm*)comm;
    if (!d || d->magic != DP_MAGIC) return mi355_fail(MI355_ERR_ARG, "dp_comm_world: not a communicator");
    return d->world;
}

int mi355_dp_comm_destroy(void* comm) {
    DpComm* d = (DpComm*)comm;
    if (!d || d->magic != DP_MAGIC) return mi355_fail(MI355_ERR_ARG, "dp_comm_destroy: not a communicator");
    const Rccl* r = rccl();
    int rc = 0;
    if (r) rc = r->comm_destroy(d->comm);
    d->magic = 0;
    delete d;
    if (rc != 0) return rccl_fail(r, "dp_comm_destroy", rc);
    return MI355_OK;
}

// In-place sum over the ranks of `count` elements at `buf`, enqueued on `stream`: dtype 0 = bf16, 1 = fp32.
int mi355_dp_allreduce_sum(void* comm, void* buf, long long count, int dtype, void* stream) {
    DpComm* d = (DpComm*)comm;
    if (!d || d->magic != DP_MAGIC) return mi355_fail(MI355_ERR_ARG, "dp_allreduce_sum: not a communicator");
    if (!buf || count < 0 || (dtype != 0 && dtype != 1)) return mi355_fail(MI355_ERR_ARG, "dp_allreduce_sum: bad arguments");
    if (count == 0) return MI355_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != d->device)
        return mi355_fail(MI355_ERR_ARG, "dp_allreduce_sum: communicator belongs to device %d, current device is %d", d->device, dev);
    const Rccl* r = rccl();
    if (!r) return mi355_fail(MI355_ERR_UNSUPPORTED, "dp: librccl.so.1 not found or incomplete");
    const int rc = r->all_reduce(buf, buf, (size_t)count, dtype == 0 ? 9 : 7, 0, d->comm, (hipStream_t)stream);
    if (rc != 0) return rccl_fail(r, "dp_allreduce_sum", rc);
    return MI355_OK;
}

}  // extern "C"
