/* qexhip_tune.h -- measurement scaffolding, libqexhip_tune.so (NOT part of the product library libqexhip.so).
 *
 * A/B variants of the one-parity sweep and calibration kernels with known byte / flop counts, used by scratch/ and
 * profiles/pmc_workload.py to decide what goes into csrc/dslash.hip and to calibrate the rocprofv3 FETCH_SIZE /
 * WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section).  No counterpart in the reference.  The library links
 * against libqexhip.so and works on a handle created by qexhip_init. */
#ifndef QEXHIP_TUNE_H
#define QEXHIP_TUNE_H
#include <stddef.h>
#include "qexhip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* run sweep variant `variant` nrep times on scratch fields of the context's lattice; average launch time in us */
int qexhip_tune_dslash(qexhip_handle h, int variant, int swz, int nrep, double *avg_us);
/* |out|^2 of the last qexhip_tune_dslash run (all variants must agree bit for bit) */
int qexhip_tune_dslash_norm2(qexhip_handle h, double *n2);
/* mode 0: 16 B/lane streaming read (k_read16), 1: copy (k_copy16) of `mbytes` MiB; GB/s */
int qexhip_tune_stream(qexhip_handle h, int mode, size_t mbytes, int nblocks, int nrep, double *gbs);
/* fp64 FMA chains: the vector-pipe ceiling the flow stage is priced against; TFLOP/s */
int qexhip_tune_fma64(qexhip_handle h, int kind, int chains, int wps, int iters, double *tflops);
/* the Wilson-flow stage's operand gathers alone (48 matrices per 64-site tile of the resident links, into registers):
 * nw wavefronts per workgroup (divides 48), wgpc workgroups per CU, depth matrices in flight per wavefront (1, 2, 4, 6); us */
int qexhip_tune_gather(qexhip_handle h, int nw, int wgpc, int depth, int nrep, double *avg_us);
/* the same stream with rows = 2: only rows 0,1 of every matrix are gathered (6 of 9 sixteen-byte requests per lane; round 6) */
int qexhip_tune_gather_rows(qexhip_handle h, int nw, int wgpc, int depth, int rows, int nrep, double *avg_us);
/* the same operand stream with the links read in a brick tile shape (8 x 4 x 4 x 1 sites per tile position): the round-5 tile-shape
 * experiment (profiles/r05_tile_shape.md); lattices with 8 | X, 4 | Y, 4 | Z */
int qexhip_tune_gather_brick(qexhip_handle h, int nw, int wgpc, int depth, int nrep, double *avg_us);
#ifdef __cplusplus
}
#endif
#endif
