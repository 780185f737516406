/* examples/c_abi_sharded.c -- several devices, one call, from plain C99.
 * Builds one CubicSpline interpolator, makes one replica per visible device (ndi_interp1d_clone: the device-resident
 * tables are copied device to device; on a 1-GPU box two replicas share device 0), hands all handles and the whole
 * flattened query array to ndi_interp1d_eval_sharded and checks the result against the single-handle call -- values
 * and the reference's first-error behaviour over the whole batch (src/interp1d/mod.rs:326-343).
 *
 *   gcc -std=c99 -Wall -pedantic examples/c_abi_sharded.c -Iinclude -Lndarray-interp_amd -lndinterp_hip
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndinterp.h"

#define N 64
#define L 8
#define Q 1001
#define MAX_SHARDS 8

int main(void) {
  static double x[N], y[N * L], q[Q], whole[Q * L], sharded[Q * L];
  ndi_interp1d_desc d;
  ndi_interp1d* h[MAX_SHARDS];
  const ndi_interp1d* hc[MAX_SHARDS];
  ndi_shard_io io[MAX_SHARDS];
  ndi_eval_opts opts;
  ndi_oob_info info;
  ndi_status st;
  int n_dev = ndi_device_count(), n_shards, i;

  if (n_dev <= 0) { fprintf(stderr, "no HIP device available; this library has no CPU fallback\n"); return 1; }
  n_shards = n_dev >= 2 ? (n_dev > MAX_SHARDS ? MAX_SHARDS : n_dev) : 2;
  for (i = 0; i < N; ++i) x[i] = 0.25 * i + 0.01 * (i % 3);
  for (i = 0; i < N * L; ++i) y[i] = (double)((i * 37) % 101) / 101.0;
  for (i = 0; i < Q; ++i) q[i] = x[0] + (x[N - 1] - x[0]) * (double)((i * 7919) % Q) / (double)Q;

  memset(&d, 0, sizeof d);
  d.dtype = NDI_F64; d.strategy = NDI_CUBIC_SPLINE; d.device = 0;
  d.n = N; d.lanes = L; d.x_len = N; d.x = x; d.data = y; d.memspace = NDI_MEM_HOST; d.validate = 1;
  st = ndi_interp1d_create(&d, &h[0]);
  if (st != NDI_OK) { fprintf(stderr, "create: %d %s\n", (int)st, ndi_last_error_string()); return 2; }
  for (i = 1; i < n_shards; ++i) {   /* replicas: knots / data / spline tables copied device to device */
    st = ndi_interp1d_clone(h[0], n_dev >= 2 ? i : 0, &h[i]);
    if (st != NDI_OK) { fprintf(stderr, "clone: %d %s\n", (int)st, ndi_last_error_string()); return 3; }
  }
  memset(&opts, 0, sizeof opts);   /* host queries, host output */
  memset(&info, 0, sizeof info);
  st = ndi_interp1d_eval(h[0], q, Q, whole, L, &opts, &info);
  if (st != NDI_OK) { fprintf(stderr, "eval: %d %s\n", (int)st, ndi_last_error_string()); return 4; }

  memset(io, 0, sizeof io);
  for (i = 0; i < n_shards; ++i) {   /* shard i writes its rows of the one host output array */
    uint64_t lo, hi;
    ndi_shard_bounds(Q, (uint32_t)i, (uint32_t)n_shards, &lo, &hi);
    io[i].out = sharded + lo * L;
    hc[i] = h[i];
  }
  st = ndi_interp1d_eval_sharded(hc, (uint32_t)n_shards, q, Q, io, L, &opts, &info);
  if (st != NDI_OK) { fprintf(stderr, "eval_sharded: %d %s\n", (int)st, ndi_last_error_string()); return 5; }
  if (memcmp(whole, sharded, sizeof whole) != 0) { fprintf(stderr, "sharded result differs\n"); return 6; }

  /* failures in the last and in the first shard: the lowest flat index of the whole batch is reported, rows before
   * it are written, later rows stay untouched */
  q[Q - 3] = 1e9; q[5] = -1e9;
  for (i = 0; i < Q * L; ++i) sharded[i] = -7.0;
  st = ndi_interp1d_eval_sharded(hc, (uint32_t)n_shards, q, Q, io, L, &opts, &info);
  if (st != NDI_OUT_OF_BOUNDS || info.index != 5 || info.value != -1e9 || info.axis != 0) {
    fprintf(stderr, "first error: status %d index %lu\n", (int)st, (unsigned long)info.index);
    return 7;
  }
  if (memcmp(whole, sharded, 5 * L * sizeof(double)) != 0) return 8;
  for (i = 5 * L; i < Q * L; ++i)
    if (sharded[i] != -7.0) return 9;
  printf("%d shards on %d device(s): sharded == single-handle result; first error at flat index %lu (%s)\n", n_shards,
         n_dev, (unsigned long)info.index, ndi_last_error_string());
  for (i = 0; i < n_shards; ++i) ndi_interp1d_destroy(h[i]);
  return 0;
}
