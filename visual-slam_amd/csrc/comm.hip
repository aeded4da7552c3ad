// comm.hip -- the one collective of the sharded batched mode behind the C-ABI: the final gather of map points to one rank
// over RCCL / xGMI (SURVEY.md 8b proposal mo_gather_map_points, 8e).  Frames are independent and pairs are owned by the rank
// that holds their frames (vslam_amd/sharding.py), so nothing else is ever exchanged.
//
// RCCL is bound at run time (dlopen): a process that never calls these entry points needs no RCCL, and a host program that
// already carries one (PyTorch-ROCm bundles its own librccl) gets THAT copy, not a second one next to a second HIP runtime.
// The communicator is created from a 128-byte id that rank 0 makes (mo_comm_unique_id) and the host program hands to the other
// ranks by whatever channel it has (torch.distributed store, MPI, a file).
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace {

typedef struct { char internal[128]; } nccl_id;
typedef void* nccl_comm;
enum { NCCL_INT32 = 2, NCCL_FLOAT32 = 7 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.lib) return r;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)  // a copy the process already loaded (PyTorch's) wins
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
        if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) return r;
#define BIND(f, sym) r.f = (decltype(r.f))dlsym(r.lib, sym)
    BIND(GetUniqueId, "ncclGetUniqueId"); BIND(CommInitRank, "ncclCommInitRank"); BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllGather, "ncclAllGather"); BIND(Send, "ncclSend"); BIND(Recv, "ncclRecv");
    BIND(GroupStart, "ncclGroupStart"); BIND(GroupEnd, "ncclGroupEnd"); BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
}

int nccl_fail(mo_ctx* c, const char* what, int rc) {
    Rccl& r = rccl();
    return mo_fail(c, MO_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

__global__ void k_store_i32(int32_t* p, int32_t v) { *p = v; }

}  // namespace

extern "C" int mo_comm_unique_id(uint8_t id[128]) {
    Rccl& r = rccl();
    if (!r.ok || !id) return MO_ERR_UNSUPPORTED;
    nccl_id u;
    if (r.GetUniqueId(&u) != 0) return MO_ERR_HIP;
    std::memcpy(id, &u, sizeof(u));
    return MO_OK;
}

extern "C" int mo_comm_init(mo_ctx* c, const uint8_t id[128], int rank, int world) {
    if (!c) return MO_ERR_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return mo_fail(c, MO_ERR_ARG, "mo_comm_init: bad id / rank / world");
    Rccl& r = rccl();
    if (!r.ok) return mo_fail(c, MO_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) { r.CommDestroy((nccl_comm)c->comm); c->comm = nullptr; }
    nccl_id u;
    std::memcpy(&u, id, sizeof(u));
    nccl_comm comm = nullptr;
    int rc = r.CommInitRank(&comm, world, u, rank);
    if (rc != 0) return nccl_fail(c, "ncclCommInitRank", rc);
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return MO_OK;
}

extern "C" int mo_comm_destroy(mo_ctx* c) {
    if (!c) return MO_ERR_ARG;
    if (c->comm) {
        hipSetDevice(c->device);
        hipStreamSynchronize(c->stream);
        rccl().CommDestroy((nccl_comm)c->comm);
        c->comm = nullptr;
    }
    return MO_OK;
}

// d_local [rows_max][cap][3] f32 (rows_local <= rows_max valid, the rest padding); on root d_all [world][rows_max][cap][3];
// d_rows_all [world] int32 on EVERY rank (device).  Enqueued on the context stream; no host synchronisation.
extern "C" int mo_gather_map_points(mo_ctx* c, const float* d_local, int rows_local, int rows_max, int cap, int root, float* d_all,
                                    int32_t* d_rows_all) {
    if (!c) return MO_ERR_ARG;
    if (!c->comm) return mo_fail(c, MO_ERR_ARG, "mo_gather_map_points: call mo_comm_init first");
    if (!d_local || !d_rows_all || rows_local < 0 || rows_local > rows_max || cap < 1 || root < 0 || root >= c->comm_world)
        return mo_fail(c, MO_ERR_ARG, "mo_gather_map_points: bad argument");
    if (c->comm_rank == root && !d_all) return mo_fail(c, MO_ERR_ARG, "mo_gather_map_points: root needs d_all");
    Rccl& r = rccl();
    HIPCHK(c, hipSetDevice(c->device));
    nccl_comm comm = (nccl_comm)c->comm;
    // 1. per-rank row counts (one int each): all-gather, so that every rank can index the result
    // (a word of its own: the gather may run on a side stream beside the next call's kernels, which use the context's other buffers;
    //  the count travels as a kernel argument: an async copy from this function's stack would outlive the variable)
    int rc;
    if (!c->d_comm_cnt) HIPCHK(c, hipMalloc((void**)&c->d_comm_cnt, 256));
    hipLaunchKernelGGL(k_store_i32, dim3(1), dim3(1), 0, c->stream, c->d_comm_cnt, (int32_t)rows_local);
    HIPCHK(c, hipGetLastError());
    if ((rc = r.AllGather(c->d_comm_cnt, d_rows_all, 1, NCCL_INT32, comm, c->stream)) != 0) return nccl_fail(c, "ncclAllGather", rc);
    // 2. padded point slabs to the root: one send per rank, `world` receives on the root, one group (point-to-point over xGMI;
    //    the payload is MBs, so this is latency-bound and a ring collective would buy nothing)
    const size_t slab = (size_t)rows_max * cap * 3;
    if ((rc = r.GroupStart()) != 0) return nccl_fail(c, "ncclGroupStart", rc);
    // an error inside the group must still close it: an open group would swallow the communicator's next collective
    const char* what = nullptr;
    if (c->comm_rank == root)
        for (int p = 0; p < c->comm_world && rc == 0; p++)
            if ((rc = r.Recv(d_all + (size_t)p * slab, slab, NCCL_FLOAT32, p, comm, c->stream)) != 0) what = "ncclRecv";
    if (rc == 0 && (rc = r.Send(d_local, slab, NCCL_FLOAT32, root, comm, c->stream)) != 0) what = "ncclSend";
    const int rc_end = r.GroupEnd();
    if (rc != 0) return nccl_fail(c, what, rc);
    if (rc_end != 0) return nccl_fail(c, "ncclGroupEnd", rc_end);
    return MO_OK;
}
