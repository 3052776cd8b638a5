// The exchange step of the attribute-sharded path as a C ABI: safe_allgather_cols plays the part of the
// np.concatenate(combined_nes, axis=1) of the reference's multiprocessing driver (safepy/safe.py:1355) for hosts that
// do not bring their own collective library -- a thin wrapper over RCCL's ncclAllGather on the context's stream.
// RCCL is loaded on first use (dlopen), so the library itself has no link-time dependency on it; a process that
// already holds a copy (PyTorch-ROCm ships one) shares that copy.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;       // ncclUniqueId
typedef void *rccl_comm;

struct Rccl {
    void *lib = nullptr;
    int (*get_unique_id)(rccl_unique_id *) = nullptr;
    int (*comm_init_rank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
    int (*all_gather)(const void *, void *, size_t, int, rccl_comm, hipStream_t) = nullptr;
    int (*comm_destroy)(rccl_comm) = nullptr;
    const char *(*get_error_string)(int) = nullptr;
};

int load_rccl(Rccl **out) {
    static Rccl r;
    static int state = 0;                                    // 0 = not tried, 1 = loaded, -1 = unavailable
    static char why[256] = "missing symbols";                // the loader's reason, kept from the one attempt
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names)                         // a copy the process already holds first
            if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
        // the copy the host language's GPU framework ships, if the binding named one (safepy_amd/_lib.py: PyTorch-ROCm's)
        if (const char *preferred = getenv("SAFE_HIP_RCCL_PATH"))
            if (!r.lib && preferred[0]) r.lib = dlopen(preferred, RTLD_NOW | RTLD_GLOBAL);
        for (size_t i = 0; !r.lib && i < sizeof(names) / sizeof(names[0]); ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) {
            r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.lib, "ncclAllGather"));
            r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.get_error_string = reinterpret_cast<decltype(r.get_error_string)>(dlsym(r.lib, "ncclGetErrorString"));
        }
        if (!r.lib) {
            const char *msg = dlerror();                     // (dlerror() clears the message: read it once)
            if (msg) snprintf(why, sizeof(why), "%s", msg);
        }
        state = (r.lib && r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy) ? 1 : -1;
    });
    if (state < 0) {
        safe_set_error("RCCL is not available: librccl.so could not be loaded (%s)", why);
        return SAFE_E_UNSUPPORTED;
    }
    *out = &r;
    return SAFE_OK;
}

int rccl_fail(const Rccl *r, const char *what, int code) {
    safe_set_error("%s failed: %s (RCCL error %d)", what, r->get_error_string ? r->get_error_string(code) : "?", code);
    return SAFE_E_HIP;
}

}  // namespace

struct safe_comm {
    safe_ctx *ctx = nullptr;
    Rccl *rccl = nullptr;
    rccl_comm comm = nullptr;
    int world = 0, rank = 0;
};

extern "C" {

int safe_comm_unique_id(char *id_out, size_t id_len) {
    SAFE_REQUIRE(id_out && id_len >= SAFE_COMM_ID_BYTES, "safe_comm_unique_id: the id buffer must hold %d bytes", SAFE_COMM_ID_BYTES);
    Rccl *r = nullptr;
    SAFE_TRY(load_rccl(&r));
    rccl_unique_id id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, SAFE_COMM_ID_BYTES);
    return SAFE_OK;
}

int safe_comm_create(safe_ctx *ctx, int world_size, int rank, const char *id, size_t id_len, safe_comm **out) {
    SAFE_REQUIRE(ctx && id && out, "safe_comm_create: NULL argument");
    SAFE_REQUIRE(id_len >= SAFE_COMM_ID_BYTES, "safe_comm_create: the id must be the %d bytes of safe_comm_unique_id", SAFE_COMM_ID_BYTES);
    SAFE_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "safe_comm_create: rank %d of %d", rank, world_size);
    *out = nullptr;
    Rccl *r = nullptr;
    SAFE_TRY(load_rccl(&r));
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    rccl_unique_id uid;
    memcpy(uid.internal, id, SAFE_COMM_ID_BYTES);
    rccl_comm c = nullptr;
    const int rc = r->comm_init_rank(&c, world_size, uid, rank);
    if (rc != 0) return rccl_fail(r, "ncclCommInitRank", rc);
    safe_comm *comm = new safe_comm();
    comm->ctx = ctx;
    comm->rccl = r;
    comm->comm = c;
    comm->world = world_size;
    comm->rank = rank;
    *out = comm;
    return SAFE_OK;
}

int safe_comm_destroy(safe_comm *comm) {
    if (!comm) return SAFE_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)safe_stream_sync(comm->ctx->stream);
    if (comm->comm) (void)comm->rccl->comm_destroy(comm->comm);
    delete comm;
    return SAFE_OK;
}

int safe_allgather_cols(safe_comm *comm, const void *local_dev, size_t bytes_per_rank, void *all_dev) {
    SAFE_REQUIRE(comm && local_dev && all_dev, "safe_allgather_cols: NULL argument");
    SAFE_HIP_CHECK(hipSetDevice(comm->ctx->device));
    if (bytes_per_rank == 0) return SAFE_OK;
    const int rc = comm->rccl->all_gather(local_dev, all_dev, bytes_per_rank, /* ncclInt8 */ 0, comm->comm, comm->ctx->stream);
    if (rc != 0) return rccl_fail(comm->rccl, "ncclAllGather", rc);
    return SAFE_OK;
}

}  // extern "C"
