// The path's one exchange step through the C ABI: in-place SUM all-reduce of an output grid over
// RCCL (xGMI inside a node), so that a caller of libevk.so that does not use torch can shard events across GPUs too.
// Every accumulator of the path is a sum over events (image.py:95,111-114,132-135: index_put_(accumulate=True);
// image.py:37: np.bincount), hence partial grids of disjoint event shards add up to the result; int32 keeps the integer
// event image bit-exact.
//
// librccl is bound at run time (dlopen + dlsym), not at link time: a process that already holds an RCCL -- a PyTorch
// program has torch/lib/librccl.so loaded -- must use THAT copy (two RCCL instances in one process do not share their
// device state), and a single-GPU user of libevk.so needs no RCCL at all.  EVK_RCCL_PATH overrides the library name.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

#include "evk_common.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        const char *env = getenv("EVK_RCCL_PATH");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            // RTLD_NOLOAD first: reuse the copy the process already has (same soname), whatever path it came from
            x.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (!x.handle) x.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
        x.get_unique_id = (decltype(x.get_unique_id))dlsym(x.handle, "ncclGetUniqueId");
        x.comm_init_rank = (decltype(x.comm_init_rank))dlsym(x.handle, "ncclCommInitRank");
        x.comm_destroy = (decltype(x.comm_destroy))dlsym(x.handle, "ncclCommDestroy");
        x.all_reduce = (decltype(x.all_reduce))dlsym(x.handle, "ncclAllReduce");
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_reduce;
        return x;
    }();
    return r;
}

int rc_of(ncclResult_t r) { return r == ncclSuccess ? EVK_OK : EVK_ECOMM; }

}  // namespace

extern "C" int evk_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int evk_comm_unique_id(void *host_id) {
    if (!host_id) return EVK_EINVAL;
    if (!rccl().ok) return EVK_ECOMM;
    ncclUniqueId id;
    const int rc = rc_of(rccl().get_unique_id(&id));
    if (rc == EVK_OK) memcpy(host_id, &id, sizeof(id));
    return rc;
}

extern "C" int evk_comm_init(const void *host_id, int rank, int world, void **comm_out) {
    if (!host_id || !comm_out || world < 1 || rank < 0 || rank >= world) return EVK_EINVAL;
    if (!rccl().ok) return EVK_ECOMM;
    ncclUniqueId id;
    memcpy(&id, host_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const int rc = rc_of(rccl().comm_init_rank(&comm, world, id, rank));
    *comm_out = rc == EVK_OK ? (void *)comm : nullptr;
    return rc;
}

extern "C" int evk_comm_destroy(void *comm) {
    if (!comm) return EVK_EINVAL;
    if (!rccl().ok) return EVK_ECOMM;
    return rc_of(rccl().comm_destroy((ncclComm_t)comm));
}

extern "C" int evk_allreduce_f32(float *buf, int64_t count, void *comm, void *stream) {
    if (!buf || count < 0 || !comm) return EVK_EINVAL;
    if (!rccl().ok) return EVK_ECOMM;
    return rc_of(rccl().all_reduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int evk_allreduce_i32(int32_t *buf, int64_t count, void *comm, void *stream) {
    if (!buf || count < 0 || !comm) return EVK_EINVAL;
    if (!rccl().ok) return EVK_ECOMM;
    return rc_of(rccl().all_reduce(buf, buf, (size_t)count, ncclInt32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}
