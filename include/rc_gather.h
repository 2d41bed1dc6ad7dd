/*
 * rc_gather.h -- C ABI of librc_gather.so: row-band assembly of one frame over RCCL (xGMI).
 *
 * The reference renders on a single device and has no counterpart (one launch on the default stream,
 * RayCasting/RayCasting/kernel.cu:123; device selection RayTracingKernel.cpp:47-82).  Here a frame is
 * cut into `world` row bands, every rank (one process per GPU) renders its band with
 * rc_render(rc_opts.y_offset / full_height, include/rc_abi.h) and the bands are assembled on rank 0:
 * the root receives every peer's rows straight into their place in the full frame, each peer sends
 * once -- point to point, one grouped RCCL operation, each transfer on that pair's own xGMI link.
 * No all-gather (it would move `world` times the data).
 *
 * A band may be rendered and sent in `parts` pieces: the transfer of piece k runs on the library's own
 * stream behind an event recorded on the caller's stream, so it travels while piece k+1 renders.
 *
 * Status: the exchange between two or more ranks has not run on hardware yet (the build had one GPU per box);
 * with one rank, where the assembly is a device-to-device copy, it has (tests/test_parity_gpu.py).
 *
 * Conventions as in rc_abi.h: 0 on success, negative on failure (-(hipError_t), RC_GATHER_ERCCL - the
 * ncclResult_t, or RC_GATHER_EINVAL); never exits, never throws; rc_gather_last_error() describes the
 * calling thread's last failure.  The library links librccl; librc_hip.so does not.
 */
#ifndef RC_GATHER_H
#define RC_GATHER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_GATHER_EINVAL (-100001)
#define RC_GATHER_ERCCL (-200000)
#define RC_GATHER_ID_BYTES 128

typedef struct rc_gather rc_gather;

/* Rows [*begin, *end) of `rank` among `world` ranks: band heights differ by at most one row, the first
 * height % world ranks get the longer bands.  Pure host arithmetic (no GPU needed). */
void rc_band_rows(int height, int world, int rank, int *begin, int *end);
/* Piece k of `parts` nearly equal pieces of rows [begin, end) (the first (end - begin) % parts pieces get
 * the extra row; parts is clamped to the number of rows).  *sb == *se for a piece that does not exist. */
void rc_sub_band(int begin, int end, int parts, int k, int *sb, int *se);

/* Rank 0 makes the 128-byte identifier every rank passes to rc_gather_create (ncclGetUniqueId); the
 * caller carries it to the other processes (a file, a socket, torch.distributed, MPI ...). */
int rc_gather_unique_id(void *id);
/* Joins the communicator on the CURRENT device (ncclCommInitRank: blocks until all `world` ranks have
 * called it).  world == 1 is allowed: the assembly is then a device-to-device copy. */
int rc_gather_create(rc_gather **g, int world, int rank, const void *id);
/* Optional, once per frame BEFORE the frame's renders are enqueued: rank 0 posts the receives of all `parts`
 * pieces of every peer now, ordered only behind the work already on `stream` (whatever still reads the previous
 * contents of `full`), so the peers' rows arrive while rank 0 renders its own band.  Without this call rank 0
 * posts the receives of piece k inside rc_gather_piece(k), behind its own render of that piece.  No-op on the
 * other ranks and with one rank.  The same parts / height / row_bytes as the rc_gather_piece calls that follow. */
int rc_gather_begin(rc_gather *g, int parts, void *full, size_t row_bytes, int height, void *stream);
/* Piece k of this rank's band goes to its rows of the full frame on rank 0.
 *   band       this rank's band: rows rc_band_rows(height, world, rank), contiguous, row_bytes each (DEVICE)
 *   full       rank 0: the whole frame, height x row_bytes, contiguous (DEVICE); ignored on other ranks
 *   stream     the stream piece k was rendered on; the transfer starts when that work has finished
 * Every rank calls it for k = 0 .. parts-1 in order, with the same parts / height / row_bytes.
 * Enqueue only; rc_gather_wait orders the caller's stream behind the transfers. */
int rc_gather_piece(rc_gather *g, int k, int parts, const void *band, void *full, size_t row_bytes,
                    int height, void *stream);
/* Work enqueued on `stream` after this call starts when all pieces handed over so far have arrived
 * (root) / have been sent (peers).  Does not block the host. */
int rc_gather_wait(rc_gather *g, void *stream);
int rc_gather_destroy(rc_gather *g);

const char *rc_gather_last_error(void);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* RC_GATHER_H */
