// Data-parallel exchange group over RCCL (xGMI), the C-ABI "comm" row of SURVEY.md 8(b).
// Replaces (reference file:line): dist.init_process_group("nccl") distributed_image_translation.py:31-38, the
// DistributedDataParallel gradient all-reduce :401-404,513-518, dist.barrier() :398,573-574 and
// dist.destroy_process_group() :42-46.
//
// One communicator per process (one process per GPU).  Collectives are enqueued on the CALLER's HIP stream
// (one hop: no library-internal stream), in place, fp32 sum; the 1/W of DDP's mean rides in the Adam kernel.
// The library holds nothing but the communicator handle between dg_dp_init and dg_dp_destroy.
// RCCL is bound at run time (dlopen "librccl.so.1"; a process that already loaded RCCL -- torch does -- gets that
// same instance), so libdiscogan_hip.so has no link-time dependency on it and single-GPU users never load it.
#include "dg_common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} R;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0;

int bind_rccl() {
    if (R.handle) return DG_OK;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return dg_fail(DG_ERR_HIP, "dg_dp: cannot load librccl.so.1: %s", dlerror());
#define BIND(field, sym)                                                                   \
    do {                                                                                   \
        *(void**)(&R.field) = dlsym(h, sym);                                               \
        if (!R.field) { dlclose(h); return dg_fail(DG_ERR_HIP, "dg_dp: librccl lacks %s", sym); } \
    } while (0)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(CommCount, "ncclCommCount");
    BIND(AllReduce, "ncclAllReduce");
    BIND(Broadcast, "ncclBroadcast");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    R.handle = h;
    return DG_OK;
}
int nccl_fail(const char* who, ncclResult_t r) {
    return dg_fail(DG_ERR_HIP, "%s: RCCL error %d (%s)", who, (int)r, R.GetErrorString ? R.GetErrorString(r) : "?");
}
}  // namespace

extern "C" int dg_dp_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// Local readiness probe, nothing collective: everything that can fail on ONE rank before ncclCommInitRank is checked
// here, so the ranks can agree on the outcome before any of them enters the blocking collective.
extern "C" int dg_dp_ready(int* device_out) {
    if (g_comm) return dg_fail(DG_ERR_INVALID, "dg_dp_ready: a communicator already exists (call dg_dp_destroy first)");
    int rc = bind_rccl();
    if (rc) return rc;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return dg_fail(DG_ERR_HIP, "dg_dp_ready: no current HIP device: %s", hipGetErrorString(e));
    if (device_out) *device_out = dev;
    return DG_OK;
}

extern "C" int dg_dp_get_unique_id(void* id_out, size_t bytes) {
    DG_CHECK_ARG(id_out && bytes >= NCCL_UNIQUE_ID_BYTES, "dg_dp_get_unique_id: need a %d-byte host buffer", NCCL_UNIQUE_ID_BYTES);
    int rc = bind_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, NCCL_UNIQUE_ID_BYTES);
    return DG_OK;
}

extern "C" int dg_dp_init(int rank, int world, const void* unique_id, size_t bytes) {
    DG_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "dg_dp_init: bad rank/world (%d/%d)", rank, world);
    DG_CHECK_ARG(unique_id && bytes >= NCCL_UNIQUE_ID_BYTES, "dg_dp_init: need the %d-byte id from dg_dp_get_unique_id", NCCL_UNIQUE_ID_BYTES);
    if (g_comm) return dg_fail(DG_ERR_INVALID, "dg_dp_init: a communicator already exists (call dg_dp_destroy first)");
    int rc = bind_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    ncclResult_t r = R.CommInitRank(&c, world, id, rank);     // collective over all ranks; uses the current HIP device
    if (r != ncclSuccess) return nccl_fail("ncclCommInitRank", r);
    g_comm = c;
    g_rank = rank;
    g_world = world;
    return DG_OK;
}

extern "C" int dg_dp_world_size(void) {
    if (!g_comm) return 0;
    int n = 0;
    if (R.CommCount(g_comm, &n) != ncclSuccess) return -1;
    return n;
}
extern "C" int dg_dp_rank(void) { return g_comm ? g_rank : -1; }

static int allreduce_f32(const char* who, float* buf, size_t n, ncclRedOp_t op, dg_stream_t stream) {
    if (!buf && n) return dg_fail(DG_ERR_INVALID, "%s: null buffer", who);
    if (!g_comm) return dg_fail(DG_ERR_INVALID, "%s: no communicator (dg_dp_init)", who);
    if (n == 0) return DG_OK;
    ncclResult_t r = R.AllReduce(buf, buf, n, ncclFloat32, op, g_comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    return DG_OK;
}
extern "C" int dg_dp_allreduce_sum(float* buf, size_t n, dg_stream_t stream) {
    return allreduce_f32("dg_dp_allreduce_sum", buf, n, ncclSum, stream);
}
extern "C" int dg_dp_allreduce_max(float* buf, size_t n, dg_stream_t stream) {
    return allreduce_f32("dg_dp_allreduce_max", buf, n, ncclMax, stream);
}

extern "C" int dg_dp_broadcast(float* buf, size_t n, int root, dg_stream_t stream) {
    DG_CHECK_ARG(buf || n == 0, "dg_dp_broadcast: null buffer");
    if (!g_comm) return dg_fail(DG_ERR_INVALID, "dg_dp_broadcast: no communicator (dg_dp_init)");
    DG_CHECK_ARG(root >= 0 && root < g_world, "dg_dp_broadcast: bad root %d", root);
    if (n == 0) return DG_OK;
    ncclResult_t r = R.Broadcast(buf, buf, n, ncclFloat32, root, g_comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail("ncclBroadcast", r);
    return DG_OK;
}

// Barrier = all-reduce of one caller-owned device float (the library allocates no device memory); the caller
// synchronises the stream to complete it on the host side.
extern "C" int dg_dp_barrier(float* scratch1, dg_stream_t stream) {
    DG_CHECK_ARG(scratch1, "dg_dp_barrier: need one device float of scratch");
    return dg_dp_allreduce_sum(scratch1, 1, stream);
}

extern "C" int dg_dp_destroy(void) {
    if (!g_comm) return DG_OK;
    ncclResult_t r = R.CommDestroy(g_comm);
    g_comm = nullptr;
    g_rank = -1;
    g_world = 0;
    if (r != ncclSuccess) return nccl_fail("ncclCommDestroy", r);
    return DG_OK;
}
