// hs_comm.hip — the ONE collective of the path in C: an RCCL all-gather of the fixed-size frame records (BASELINE config 5).
// The reference has no multi-camera exchange; north_star adds it ("RCCL all-gather over xGMI of per-frame keypoints / descriptors for
// cross-camera matching") and keeps the host in C++, so the exchange must not need Python: hs_comm_* wraps ncclGetUniqueId /
// ncclCommInitRank / ncclAllGather / ncclCommDestroy.  The caller carries the 128-byte id from rank 0 to the other ranks over whatever channel
// it already has (a file, a socket, MPI, torch.distributed) — the same contract as NCCL's own bootstrap.
// librccl is resolved with dlopen on first use: the extractor / matcher library has no link-time dependency on RCCL, a process that never
// creates a communicator never loads it, and a process that already loaded an RCCL (PyTorch ships one) shares that copy.
// The all-gather is enqueued on the caller's stream (default: the handle's own): with the extraction before it and the matcher after it on the
// SAME stream the three stages of a config-5 step are ordered without events and without touching the host.
#include "hs_internal.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>

void hs_set_error(hs_orb* h, const char* msg);       // hs_api.hip
int hs_orb_device_of(const hs_orb* h);              // hs_api.hip
hipStream_t hs_orb_stream_of(const hs_orb* h);      // hs_api.hip
void hs_orb_borrow(hs_orb* h, int delta);           // hs_api.hip: a communicator borrows the handle it was created on

// the RCCL types this file needs (rccl.h: ncclResult_t is an enum, ncclComm_t an opaque pointer, ncclUniqueId 128 opaque bytes, ncclChar = 0)
typedef int hs_nccl_result;
typedef struct ncclComm* hs_nccl_comm;
struct hs_nccl_id { char internal[HS_COMM_ID_BYTES]; };
static_assert(HS_COMM_ID_BYTES == 128, "NCCL_UNIQUE_ID_BYTES");

namespace {
struct Rccl {
    void* so = nullptr;
    hs_nccl_result (*GetUniqueId)(hs_nccl_id*) = nullptr;
    hs_nccl_result (*CommInitRank)(hs_nccl_comm*, int, hs_nccl_id, int) = nullptr;
    hs_nccl_result (*AllGather)(const void*, void*, size_t, int, hs_nccl_comm, hipStream_t) = nullptr;
    hs_nccl_result (*CommDestroy)(hs_nccl_comm) = nullptr;
    const char* (*GetErrorString)(hs_nccl_result) = nullptr;
    hs_nccl_result (*CommCount)(const hs_nccl_comm, int*) = nullptr;      // optional: what RCCL itself says about a communicator (evidence for a multi-GPU run)
    hs_nccl_result (*CommUserRank)(const hs_nccl_comm, int*) = nullptr;
    hs_nccl_result (*GetVersion)(int*) = nullptr;
    std::string why;
};
Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // WHICH copy: the one that lives next to the HIP runtime this process actually runs on.  A process may hold two ROCm stacks (PyTorch ships its own
        // libamdhip64 + librccl; whichever HIP runtime is loaded first serves everybody) and RCCL must match the runtime whose streams it is handed:
        // PyTorch's RCCL on the system's runtime failed in ncclCommInitRank ("unhandled cuda error"), the system's RCCL next to PyTorch's runtime
        // is the same mix the other way round.  So: the directory of the loaded libamdhip64 first, then the usual names.
        std::string near1, near2;
        {
            Dl_info info;
            if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
                std::string dir(info.dli_fname);
                const size_t slash = dir.rfind('/');
                if (slash != std::string::npos) { dir.resize(slash + 1); near1 = dir + "librccl.so.1"; near2 = dir + "librccl.so"; }
            }
        }
        const char* names[] = { near1.c_str(), near2.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        // RTLD_LOCAL: a process may hold ANOTHER copy of RCCL (PyTorch ships its own and loads it by path; a copy that is already resident under the
        // same soname is simply reused).  With RTLD_GLOBAL this copy's symbols interposed on a PyTorch imported later and the process died in the static
        // destructors at exit ("double free or corruption")
        for (const char* n : names) { if (!*n) continue; r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.so) break; }
        if (!r.so) { const char* e = dlerror(); r.why = std::string("librccl not found: ") + (e ? e : ""); return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.so, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.so, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.so, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.so, "ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.so, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.so, "ncclCommUserRank"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.so, "ncclGetVersion"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { r.why = "librccl lacks an expected symbol"; r.so = nullptr; }
    });
    return r;
}
std::string nccl_text(const Rccl& r, const char* what, hs_nccl_result rc)
{
    return std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
}
}  // namespace

struct hs_comm {
    hs_orb* h = nullptr;
    hs_nccl_comm comm = nullptr;
    int world = 0, rank = 0;
    std::string err;
};

extern "C" {

// non-collective probe: can this process create a communicator at all (librccl loads and has the entry points)?  A multi-rank caller asks every
// rank BEFORE the first hs_comm_create — which blocks in ncclCommInitRank until all ranks arrive — and falls back to its own exchange when any
// rank says no (bench.py --config c5 does).
int hs_comm_available(void)
{
    Rccl& r = rccl();
    return r.so ? HS_OK : HS_ERR_NO_DEVICE;
}
const char* hs_comm_unavailable_reason(void) { Rccl& r = rccl(); return r.so ? "" : r.why.c_str(); }

int hs_comm_get_unique_id(uint8_t* id)
{
    if (!id) return HS_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.so) return HS_ERR_NO_DEVICE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); return HS_ERR_NO_DEVICE; }
    hs_nccl_id u;
    if (r.GetUniqueId(&u) != 0) return HS_ERR_HIP;
    memcpy(id, u.internal, HS_COMM_ID_BYTES);
    return HS_OK;
}

int hs_comm_create(hs_orb* h, const uint8_t* id, int world, int rank, hs_comm** out)
{
    if (!h || !id || !out || world < 1 || rank < 0 || rank >= world) return HS_ERR_INVALID;
    *out = nullptr;
    Rccl& r = rccl();
    if (!r.so) { hs_set_error(h, r.why.c_str()); return HS_ERR_NO_DEVICE; }
    if (hipSetDevice(hs_orb_device_of(h)) != hipSuccess) { (void)hipGetLastError(); hs_set_error(h, "hipSetDevice failed"); return HS_ERR_HIP; }
    hs_nccl_id u;
    memcpy(u.internal, id, HS_COMM_ID_BYTES);
    hs_comm* c = new hs_comm();
    c->h = h; c->world = world; c->rank = rank;
    const hs_nccl_result rc = r.CommInitRank(&c->comm, world, u, rank);      // blocks until all `world` ranks have called it
    if (rc != 0) { hs_set_error(h, nccl_text(r, "ncclCommInitRank", rc).c_str()); delete c; return HS_ERR_HIP; }
    hs_orb_borrow(h, +1);                                   // hs_orb_destroy(h) is deferred until this communicator is gone
    *out = c;
    return HS_OK;
}

void hs_comm_destroy(hs_comm* c)
{
    if (!c) return;
    Rccl& r = rccl();
    if (c->comm && r.so) { hipSetDevice(hs_orb_device_of(c->h)); r.CommDestroy(c->comm); }
    hs_orb* const h = c->h;
    delete c;
    hs_orb_borrow(h, -1);                                   // completes a deferred hs_orb_destroy when this was the last borrower
}

// What RCCL ITSELF reports (not what the caller passed in): ncclCommCount / ncclCommUserRank of the live communicator, ncclGetVersion of the library
// that was loaded.  -1 = unknown (no communicator, librccl not loaded, or the entry point is missing).  A multi-GPU run prints these, so that the
// first run on real hardware leaves evidence of how many ranks the collective really spanned.
int hs_comm_rccl_ranks(const hs_comm* c)
{
    Rccl& r = rccl();
    int n = -1;
    if (!c || !c->comm || !r.so || !r.CommCount || r.CommCount(c->comm, &n) != 0) return -1;
    return n;
}
int hs_comm_rccl_rank(const hs_comm* c)
{
    Rccl& r = rccl();
    int n = -1;
    if (!c || !c->comm || !r.so || !r.CommUserRank || r.CommUserRank(c->comm, &n) != 0) return -1;
    return n;
}
int hs_comm_rccl_version(void)
{
    Rccl& r = rccl();
    int v = -1;
    if (!r.so || !r.GetVersion || r.GetVersion(&v) != 0) return -1;
    return v;
}

int hs_comm_world(const hs_comm* c) { return c ? c->world : 0; }
int hs_comm_rank(const hs_comm* c) { return c ? c->rank : -1; }
const char* hs_comm_last_error(const hs_comm* c) { return c ? c->err.c_str() : "null communicator"; }

int hs_comm_allgather_records(hs_comm* c, const void* d_record, void* d_gathered, size_t record_bytes, void* stream)
{
    if (!c) return HS_ERR_INVALID;
    if (!d_record || !d_gathered || record_bytes == 0) { c->err = "bad argument"; return HS_ERR_INVALID; }
    Rccl& r = rccl();
    if (hipSetDevice(hs_orb_device_of(c->h)) != hipSuccess) { (void)hipGetLastError(); c->err = "hipSetDevice failed"; return HS_ERR_HIP; }
    hipStream_t s = stream ? (hipStream_t)stream : hs_orb_stream_of(c->h);
    // ncclChar = 0; in place when d_record == d_gathered + rank * record_bytes (NCCL's in-place all-gather convention)
    const hs_nccl_result rc = r.AllGather(d_record, d_gathered, record_bytes, 0, c->comm, s);
    if (rc != 0) { c->err = nccl_text(r, "ncclAllGather", rc); return HS_ERR_HIP; }
    return HS_OK;
}

}  // extern "C"
