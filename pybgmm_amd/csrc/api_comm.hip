// The one collective of SURVEY 8(e): the final label gather through RCCL (loaded on first use).
#include "api_internal.h"

// ------------------------------------------------------------------------------------------
// Final label gather of independent chains (include/bgmm.h): RCCL through dlopen, so that the
// library carries no link-time dependency on it (and shares the copy a host process already loaded).
// ------------------------------------------------------------------------------------------
struct Id128 { char b[128]; };                          // ncclUniqueId (passed by value)
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl g_rccl;
static std::mutex g_rccl_mutex;          // (chains driven from threads meet here on first use)

static int rccl_load() {
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (g_rccl.lib) return 0;
    // The copy that belongs to THIS library's HIP runtime (the one next to the libamdhip64 we are linked against): a
    // process may carry another RCCL built against another runtime (PyTorch bundles both), and streams and device
    // pointers of one runtime mean nothing to the other.
    void *h = nullptr;
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) h = dlopen((dir.substr(0, slash) + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(nullptr, BGMM_EDEVICE, "librccl.so.1 not found (multi-chain gather needs RCCL)");
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(h, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    r.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        return fail(nullptr, BGMM_EDEVICE, "librccl.so.1 lacks the nccl* entry points");
    g_rccl = r;
    return 0;
}

static int rccl_fail(bgmm_ctx *c, const char *what, int code) {
    std::string msg = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error");
    return fail(c, BGMM_EDEVICE, msg.c_str());
}

extern "C" int bgmm_comm_unique_id(void *id128_out) {
    if (!id128_out) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id128_out);
    return e == 0 ? 0 : rccl_fail(nullptr, "ncclGetUniqueId", e);
}

extern "C" int bgmm_comm_create(int32_t rank, int32_t world_size, const void *id128, int32_t device, void **comm_out) {
    if (!id128 || !comm_out || world_size < 1 || rank < 0 || rank >= world_size) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, BGMM_EDEVICE, "hipSetDevice failed");
    (void)hipGetLastError();                            // (RCCL reports a stale error of the thread as its own)
    Id128 id;
    memcpy(id.b, id128, sizeof(id.b));
    void *comm = nullptr;
    const int e = g_rccl.CommInitRank(&comm, world_size, id, rank);
    if (e != 0) return rccl_fail(nullptr, "ncclCommInitRank", e);
    *comm_out = comm;
    return 0;
}

extern "C" int bgmm_gather_labels(bgmm_ctx *c, void *comm, int32_t world_size, int64_t *z_all_out) {
    if (!c || !comm || !z_all_out || world_size < 1) return BGMM_EINVAL;
    SETTLE(c);
    int rc = rccl_load();
    if (rc) return rc;
    CK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->d.N;
    long long *dz = nullptr, *dall = nullptr;
    CK(c, hipMalloc((void **)&dz, sizeof(long long) * N));
    hipError_t e = hipMalloc((void **)&dall, sizeof(long long) * N * (size_t)world_size);
    if (e != hipSuccess) { (void)hipFree(dz); CK(c, e); }
    launch_labels(c->d, dz, nullptr, c->stream);
    (void)hipGetLastError();
    const int ne = g_rccl.AllGather(dz, dall, N, /* ncclInt64 */ 4, comm, c->stream);
    if (ne == 0) {
        e = hipMemcpyAsync(z_all_out, dall, sizeof(long long) * N * (size_t)world_size, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(dz); (void)hipFree(dall);
    if (ne != 0) return rccl_fail(c, "ncclAllGather", ne);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_comm_destroy(void *comm) {
    if (!comm) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    const int e = g_rccl.CommDestroy(comm);
    return e == 0 ? 0 : rccl_fail(nullptr, "ncclCommDestroy", e);
}
