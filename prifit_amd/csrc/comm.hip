// Thin RCCL export for the one exchange of the path: ONE sum all-reduce of the flat fp32 gradient
// bucket per optimizer step, on the caller's HIP stream, over xGMI.  The reference's counterpart is the gradient
// reduce inside nn.DataParallel (train_partseg_shapenet.py:248-250).
//
// RCCL is resolved at run time with dlopen / dlsym -- first the copy that is already loaded in the process (PyTorch
// ships and loads its own librccl.so), then the ROCm one -- so libprifit_hip.so has no link-time dependency on it and a
// process never ends up with two RCCL instances.  Host code only; nothing here launches a kernel of its own.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    bool ok;
    bool in_process;   // bound to an RCCL that was already loaded in the process (RTLD_NOLOAD found it)
};

const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl t;
        memset(&t, 0, sizeof(t));
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void *h = nullptr;
        for (const char *n : names)   // already loaded in this process?
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        t.in_process = h != nullptr;
        for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) return t;
        t.GetUniqueId = (decltype(t.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        t.CommInitRank = (decltype(t.CommInitRank))dlsym(h, "ncclCommInitRank");
        t.AllReduce = (decltype(t.AllReduce))dlsym(h, "ncclAllReduce");
        t.CommDestroy = (decltype(t.CommDestroy))dlsym(h, "ncclCommDestroy");
        t.ok = t.GetUniqueId && t.CommInitRank && t.AllReduce && t.CommDestroy;
        return t;
    }();
    return r;
}

}  // namespace

extern "C" {

int prifit_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int prifit_comm_in_process(void) { return rccl().ok && rccl().in_process ? 1 : 0; }

int prifit_comm_unique_id(void *out)
{
    if (!out) return PRIFIT_EINVAL;
    if (!rccl().ok) return PRIFIT_ELAUNCH;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return PRIFIT_ELAUNCH;
    memcpy(out, &id, sizeof(id));
    return 0;
}

int prifit_comm_init(void **comm, int nranks, int rank, const void *unique_id)
{
    if (!comm || !unique_id || nranks <= 0 || rank < 0 || rank >= nranks) return PRIFIT_EINVAL;
    if (!rccl().ok) return PRIFIT_ELAUNCH;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    if (rccl().CommInitRank(&c, nranks, id, rank) != ncclSuccess) return PRIFIT_ELAUNCH;
    *comm = (void *)c;
    return 0;
}

int prifit_allreduce_flat(float *buf, long long count, void *comm, void *stream)
{
    if (!buf || count <= 0 || !comm) return PRIFIT_EINVAL;
    if (!rccl().ok) return PRIFIT_ELAUNCH;
    return rccl().AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, (ncclComm_t)comm, as_stream(stream)) == ncclSuccess
               ? 0
               : PRIFIT_ELAUNCH;
}

int prifit_comm_destroy(void *comm)
{
    if (!comm) return PRIFIT_EINVAL;
    if (!rccl().ok) return PRIFIT_ELAUNCH;
    return rccl().CommDestroy((ncclComm_t)comm) == ncclSuccess ? 0 : PRIFIT_ELAUNCH;
}

}  // extern "C"
