// rc_gather.cpp -- librc_gather.so (declared in include/rc_gather.h): row-band assembly of one frame
// over RCCL.  Host code only; one process per GPU, one communicator per process.
//
// Data path per piece: peers ncclSend their rows, the root posts one ncclRecv per peer straight into
// the rows of the full frame, all inside one ncclGroupStart/End, and copies its own rows device to
// device.  Sends and the root's own copy run on the library's transfer stream behind an event recorded
// on the caller's stream, so a piece travels while the caller renders the next one.  The root's
// receives run on a second stream: posted by rc_gather_begin before the root renders anything, they
// wait only for the work that was on the caller's stream at that moment, so the peers' rows arrive
// while the root is still rendering its own.
//
// Status: the paths with more than one rank have not run on hardware yet (no multi-GPU node was
// available to the build); with one rank the assembly is the device-to-device copy, which has.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <string.h>

#include <new>

#include "rc_gather.h"

static_assert(RC_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "identifier size");

struct rc_gather {
    int world = 1, rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t xfer = nullptr;      // sends and the root's own copy
    hipStream_t recv = nullptr;      // the root's receives
    hipEvent_t ready = nullptr;      // caller's stream -> xfer / recv: work enqueued so far has finished
    hipEvent_t done = nullptr;       // xfer -> caller's stream: pieces sent / copied
    hipEvent_t arrived = nullptr;    // recv -> caller's stream: peers' pieces arrived
    int posted = 0;                  // pieces whose receives rc_gather_begin has already posted for the current frame
};

namespace {

thread_local char g_err[256] = "";

int fail_arg(const char *what)
{
    snprintf(g_err, sizeof(g_err), "invalid argument: %s", what);
    return RC_GATHER_EINVAL;
}

int fail_hip(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
    return -(int)e;
}

int fail_rccl(ncclResult_t r, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (ncclResult %d)", what, ncclGetErrorString(r), (int)r);
    return RC_GATHER_ERCCL - (int)r;
}

#define G_HIP(call)                                        \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return fail_hip(e_, #call);  \
    } while (0)
#define G_RCCL(call)                                        \
    do {                                                    \
        ncclResult_t r_ = (call);                           \
        if (r_ != ncclSuccess) return fail_rccl(r_, #call); \
    } while (0)

} // namespace

extern "C" void rc_band_rows(int height, int world, int rank, int *begin, int *end)
{
    if (world < 1) world = 1;
    const int base = height / world, rem = height % world;
    const int b = rank * base + (rank < rem ? rank : rem);
    if (begin) *begin = b;
    if (end) *end = b + base + (rank < rem ? 1 : 0);
}

extern "C" void rc_sub_band(int begin, int end, int parts, int k, int *sb, int *se)
{
    const int rows = end - begin;
    if (rows <= 0) parts = 1;
    else if (parts > rows) parts = rows;
    if (parts < 1) parts = 1;
    int b = begin, e = begin;
    if (k >= 0 && k < parts && rows > 0) {
        const int base = rows / parts, rem = rows % parts;
        b = begin + k * base + (k < rem ? k : rem);
        e = b + base + (k < rem ? 1 : 0);
    }
    if (sb) *sb = b;
    if (se) *se = e;
}

extern "C" int rc_gather_unique_id(void *id)
{
    if (!id) return fail_arg("null id");
    ncclUniqueId u;
    G_RCCL(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int rc_gather_create(rc_gather **out, int world, int rank, const void *id)
{
    if (!out || !id) return fail_arg("null pointer");
    if (world < 1 || rank < 0 || rank >= world) return fail_arg("rank outside [0, world)");
    rc_gather *g = new (std::nothrow) rc_gather;
    if (!g) return fail_arg("out of host memory");
    g->world = world;
    g->rank = rank;
    int rc = 0;
    do {
        hipError_t e = hipGetDevice(&g->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->xfer, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->recv, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->arrived, hipEventDisableTiming);
        if (e != hipSuccess) { rc = fail_hip(e, "stream / event set-up"); break; }
        ncclUniqueId u;
        memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t r = ncclCommInitRank(&g->comm, world, u, rank);
        if (r != ncclSuccess) { rc = fail_rccl(r, "ncclCommInitRank"); g->comm = nullptr; }
    } while (0);
    if (rc != 0) {
        rc_gather_destroy(g);
        return rc;
    }
    *out = g;
    return 0;
}

namespace {
// Root: one grouped receive of piece k from every peer, straight into the piece's rows of the full frame.
int post_receives(rc_gather *g, int k, int parts, char *dst, size_t row_bytes, int height, hipStream_t on)
{
    G_RCCL(ncclGroupStart());
    ncclResult_t r = ncclSuccess;
    for (int peer = 1; peer < g->world && r == ncclSuccess; peer++) {
        int pb, pe, psb, pse;
        rc_band_rows(height, g->world, peer, &pb, &pe);
        rc_sub_band(pb, pe, parts, k, &psb, &pse);
        if (pse > psb)
            r = ncclRecv(dst + (size_t)psb * row_bytes, (size_t)(pse - psb) * row_bytes, ncclUint8, peer, g->comm, on);
    }
    ncclResult_t r2 = ncclGroupEnd();
    if (r != ncclSuccess) return fail_rccl(r, "ncclRecv");
    if (r2 != ncclSuccess) return fail_rccl(r2, "ncclGroupEnd");
    return 0;
}
} // namespace

extern "C" int rc_gather_begin(rc_gather *g, int parts, void *full, size_t row_bytes, int height, void *stream)
{
    if (!g) return fail_arg("null gather");
    if (row_bytes == 0 || height <= 0 || parts < 1) return fail_arg("rc_gather_begin arguments");
    if (g->rank != 0 || g->world == 1) return 0;             // peers only send; one rank only copies
    if (!full) return fail_arg("rank 0 needs the full frame");
    int cur = -1;
    G_HIP(hipGetDevice(&cur));
    if (cur != g->device) return fail_arg("the communicator's device is not current");
    // the receives may overwrite `full` once everything enqueued on the caller's stream so far -- the
    // previous frame's readers -- has finished; they do not wait for the renders enqueued after this call
    G_HIP(hipEventRecord(g->ready, static_cast<hipStream_t>(stream)));
    G_HIP(hipStreamWaitEvent(g->recv, g->ready, 0));
    for (int k = 0; k < parts; k++) {
        int rc = post_receives(g, k, parts, static_cast<char *>(full), row_bytes, height, g->recv);
        if (rc != 0) return rc;
    }
    g->posted = parts;
    return 0;
}

extern "C" int rc_gather_piece(rc_gather *g, int k, int parts, const void *band, void *full, size_t row_bytes,
                               int height, void *stream)
{
    if (!g) return fail_arg("null gather");
    if (!band || row_bytes == 0 || height <= 0 || parts < 1 || k < 0) return fail_arg("rc_gather_piece arguments");
    if (g->rank == 0 && !full) return fail_arg("rank 0 needs the full frame");
    int cur = -1;
    G_HIP(hipGetDevice(&cur));
    if (cur != g->device) return fail_arg("the communicator's device is not current");

    int begin, end, sb, se;
    rc_band_rows(height, g->world, g->rank, &begin, &end);
    rc_sub_band(begin, end, parts, k, &sb, &se);
    const char *mine = static_cast<const char *>(band) + (size_t)(sb - begin) * row_bytes;

    // this rank's own rows move when the work already enqueued on the caller's stream (their render) is done
    G_HIP(hipEventRecord(g->ready, static_cast<hipStream_t>(stream)));
    G_HIP(hipStreamWaitEvent(g->xfer, g->ready, 0));

    if (g->rank == 0) {
        char *dst = static_cast<char *>(full);
        if (g->world > 1 && k >= g->posted) {                // not posted ahead by rc_gather_begin
            G_HIP(hipStreamWaitEvent(g->recv, g->ready, 0));
            int rc = post_receives(g, k, parts, dst, row_bytes, height, g->recv);
            if (rc != 0) return rc;
        }
        if (se > sb && dst + (size_t)sb * row_bytes != mine)
            G_HIP(hipMemcpyAsync(dst + (size_t)sb * row_bytes, mine, (size_t)(se - sb) * row_bytes, hipMemcpyDeviceToDevice, g->xfer));
    } else if (se > sb) {
        G_RCCL(ncclGroupStart());
        ncclResult_t r = ncclSend(mine, (size_t)(se - sb) * row_bytes, ncclUint8, 0, g->comm, g->xfer);
        ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess) return fail_rccl(r, "ncclSend");
        if (r2 != ncclSuccess) return fail_rccl(r2, "ncclGroupEnd");
    }
    return 0;
}

extern "C" int rc_gather_wait(rc_gather *g, void *stream)
{
    if (!g) return fail_arg("null gather");
    G_HIP(hipEventRecord(g->done, g->xfer));
    G_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->done, 0));
    if (g->rank == 0 && g->world > 1) {
        G_HIP(hipEventRecord(g->arrived, g->recv));
        G_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), g->arrived, 0));
    }
    g->posted = 0;                                           // the next frame posts its own receives
    return 0;
}

extern "C" int rc_gather_destroy(rc_gather *g)
{
    if (!g) return 0;
    int rc = 0;
    if (g->xfer) (void)hipStreamSynchronize(g->xfer);
    if (g->recv) (void)hipStreamSynchronize(g->recv);
    if (g->comm) {
        ncclResult_t r = ncclCommDestroy(g->comm);
        if (r != ncclSuccess) rc = fail_rccl(r, "ncclCommDestroy");
    }
    if (g->ready) (void)hipEventDestroy(g->ready);
    if (g->done) (void)hipEventDestroy(g->done);
    if (g->arrived) (void)hipEventDestroy(g->arrived);
    if (g->xfer) (void)hipStreamDestroy(g->xfer);
    if (g->recv) (void)hipStreamDestroy(g->recv);
    delete g;
    return rc;
}

extern "C" const char *rc_gather_last_error(void) { return g_err; }
