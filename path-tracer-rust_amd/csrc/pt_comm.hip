// pt_comm.hip — the one collective of the path behind the C ABI: the framebuffer gather over RCCL (xGMI).
//
// The reference's parallel loop fills one `pixels` vector (src/render/mod.rs:1017-1024); with one rank per GPU each
// rank holds the rows it rendered (interleaved partition: pt_config.chunk_*), and every rank gets the whole frame by ONE
// ncclAllGather of the rank buffers followed by one kernel that puts the chunks back in framebuffer order.  xGMI is
// point to point: a single all-gather of 9.4 MB (1024x768) / 805 MB (4096^2) per frame is the whole traffic, so nothing
// here is bucketed or overlapped - the frame is complete when the collective starts.
//
// librccl is loaded with dlopen on first use: libptrace_hip.so has no link-time dependency on it (nor on its headers: the
// five entry points and three types used are declared below as rccl.h declares them), and hosts that never gather do not
// load it.  WHICH librccl matters: a communicator must run on the HIP runtime this library itself is bound to - it is
// handed this library's streams and buffers.  A process may hold two HIP runtimes (a PyTorch wheel bundles its own
// libamdhip64 / libhsa-runtime64 / librccl under torch/lib; whichever of torch and this library is loaded first decides
// which runtime this library binds to, and the other copy may be mapped as well), and both RCCL copies carry the soname
// librccl.so.1, so a dlopen by soname returns whichever copy was mapped first - in a process that loaded this library
// before torch that is torch's RCCL on torch's runtime, and ncclCommInitRank fails ("unhandled cuda error").  So the
// search order is: PT_RCCL_LIB (explicit override); then librccl.so.1 / librccl.so IN THE DIRECTORY OF THE libamdhip64 THIS
// LIBRARY'S hipGetDeviceCount RESOLVES TO (dladdr), opened by full path - the sibling of our runtime, whatever else is mapped;
// then the bare sonames.
#include <dlfcn.h>
#include <link.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/ptrace.h"
#include "pt_kernels.h"

namespace pt {
void set_error(const std::string &m);
}
using namespace pt;

namespace {

// rccl.h (RCCL 2.x), the part used here
typedef int ncclResult_t;  // enum: ncclSuccess = 0
constexpr ncclResult_t ncclSuccess = 0;
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclDataType_t;  // enum: ncclFloat32 = 7
constexpr ncclDataType_t ncclFloat = 7;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
    std::string path;      // what was opened
    std::string hip_path;  // the HIP runtime this library is bound to
    int hip_runtimes = 0;  // objects named libamdhip64* mapped in the process
};

int count_hip(struct dl_phdr_info *info, size_t, void *data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) ++*(int *)data;
    return 0;
}

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string sib1, sib2;
        Dl_info di;
        if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
            r.hip_path = di.dli_fname;
            const size_t slash = r.hip_path.rfind('/');
            if (slash != std::string::npos) {
                sib1 = r.hip_path.substr(0, slash + 1) + "librccl.so.1";
                sib2 = r.hip_path.substr(0, slash + 1) + "librccl.so";
            }
        }
        dl_iterate_phdr(count_hip, &r.hip_runtimes);
        const char *names[5] = {getenv("PT_RCCL_LIB"), sib1.c_str(), sib2.c_str(), "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) {
                r.path = n;
                break;
            }
            r.why = dlerror();
        }
        if (!r.handle) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
            r.why = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            r.handle = nullptr;
        }
    });
    return r;
}

// which RCCL on which HIP runtime (appended to communicator errors: the usual cause is a mismatch of the two)
std::string rccl_where() {
    const Rccl &r = rccl();
    return " [librccl: " + r.path + "; this library's HIP runtime: " + r.hip_path + "; libamdhip64 objects mapped: " +
           std::to_string(r.hip_runtimes) + "]";
}

int need_rccl() {
    if (rccl().handle) return PT_OK;
    set_error("RCCL is not available: " + rccl().why);
    return PT_ERR_COMM;
}

#define NCCL_TRY(expr)                                                                   \
    do {                                                                                 \
        ncclResult_t r_ = (expr);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            set_error(std::string(#expr) + ": " + rccl().GetErrorString(r_));            \
            return PT_ERR_COMM;                                                          \
        }                                                                                \
    } while (0)
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                \
            return PT_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

// pixels of the band of `span` indices that fall to rank r of n (chunks of C pixels dealt round-robin)
uint32_t rank_pixels(uint64_t span, uint64_t C, uint32_t r, uint32_t n) {
    const uint64_t n_chunks = (span + C - 1) / C;
    uint64_t total = 0;
    for (uint64_t c = r; c < n_chunks; c += n) total += (c * C + C < span ? C : span - c * C);
    return (uint32_t)total;
}

}  // namespace

struct pt_comm {
    int device = 0, rank = 0, n_ranks = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    float *staging = nullptr;  // n_ranks x (largest rank buffer) floats
    size_t staging_floats = 0;
};

extern "C" {

int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes (NCCL_UNIQUE_ID_BYTES)");
    if (!id) {
        set_error("id is NULL");
        return PT_ERR_INVALID;
    }
    int rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return PT_OK;
}

int pt_comm_create(int device, int rank, int n_ranks, const uint8_t id[PT_COMM_ID_BYTES], pt_comm **out) {
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        set_error("NULL argument or rank outside [0, n_ranks)");
        return PT_ERR_INVALID;
    }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        set_error("no HIP device");
        return PT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n_dev) {
        set_error("device index out of range");
        return PT_ERR_INVALID;
    }
    int rc = need_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    pt_comm *c = new pt_comm();
    c->device = device;
    c->rank = rank;
    c->n_ranks = n_ranks;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = rccl().CommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        set_error(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r) + rccl_where());
        delete c;
        return PT_ERR_COMM;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
        (void)rccl().CommDestroy(c->comm);
        delete c;
        return PT_ERR_HIP;
    }
    *out = c;
    return PT_OK;
}

void pt_comm_destroy(pt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->staging) (void)hipFree(c->staging);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int pt_comm_gather_frame(pt_comm *c, const pt_config *cfg, const void *d_local_rgb, void *d_frame_rgb, void *hip_stream) {
    if (!c || !cfg || !d_local_rgb || !d_frame_rgb) {
        set_error("NULL argument");
        return PT_ERR_INVALID;
    }
    const uint64_t npix = (uint64_t)cfg->width * cfg->height;
    uint64_t b = cfg->idx_begin, e = cfg->idx_end;
    if (b == 0 && e == 0) e = npix;
    if (npix == 0 || npix > 0x7fffffffull || b >= e || e > npix) {
        set_error("bad frame or band");
        return PT_ERR_INVALID;
    }
    const uint64_t span = e - b;
    const uint32_t n = (uint32_t)c->n_ranks;
    const uint64_t C = n > 1 ? cfg->chunk_pixels : span;
    if (C == 0) {
        set_error("chunk_pixels must be positive when there is more than one rank");
        return PT_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    uint32_t largest = 0;
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t own = rank_pixels(span, C, r, n);
        largest = own > largest ? own : largest;
    }
    const size_t slot = (size_t)largest * 3;  // floats per rank in the staging buffer (equal counts: ncclAllGather)
    if (c->staging_floats < slot * n) {
        if (c->staging) (void)hipFree(c->staging);
        c->staging = nullptr;
        c->staging_floats = 0;
        HIP_TRY(hipMalloc((void **)&c->staging, slot * n * sizeof(float)));
        c->staging_floats = slot * n;
    }
    const uint32_t own = rank_pixels(span, C, (uint32_t)c->rank, n);
    // in place: this rank's rows go to its slot of the receive buffer, which is where ncclAllGather expects them
    float *mine = c->staging + slot * (size_t)c->rank;
    if (own) HIP_TRY(hipMemcpyAsync(mine, d_local_rgb, (size_t)own * 3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (slot) NCCL_TRY(rccl().AllGather(mine, c->staging, slot, ncclFloat, c->comm, st));
    for (uint32_t r = 0; r < n; ++r) {  // rank r's k-th chunk is the band's (k*n + r)-th chunk
        const uint32_t cnt = rank_pixels(span, C, r, n);
        if (cnt) launch_scatter_chunks(st, c->staging + slot * r, (float *)d_frame_rgb, cnt, (uint32_t)C, n, r);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return PT_OK;
}

}  // extern "C"
