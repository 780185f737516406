/* examples/c_abi_example.c -- the C ABI from plain C99 (what a cgo / JNI / Rust FFI binding sees).
 * Builds a CubicSpline interpolator from host arrays, evaluates a batch into a host buffer and prints
 * the result, then runs the same batch through the device-output ring and the resident locator; exits
 * non-zero on any error.  Link: -lndinterp_hip (needs an MI355X at run time).
 *
 *   gcc -std=c99 -Wall -pedantic examples/c_abi_example.c -Iinclude -Lndarray-interp_amd -lndinterp_hip
 */
#include <stdio.h>
#include <string.h>

#include "ndinterp.h"

/* ring consumer: called once per chunk after its kernels are enqueued; a real consumer enqueues its own work on
 * chunk->stream here (stream order then protects the slot) and returns NULL */
static void* count_rows(void* user, const ndi_ring_chunk* chunk) {
  unsigned long* rows = (unsigned long*)user;
  *rows += (unsigned long)chunk->q_count;
  return NULL;
}

int main(void) {
  /* the doctest of src/interp1d/strategies/cubic_spline.rs:62-82 */
  const double x[3] = {-1.0, 0.0, 3.0};
  const double y[3] = {0.5, 0.0, 3.0};
  double q[10], out[10];
  int i;
  ndi_interp1d_desc d;
  ndi_interp1d* h = NULL;
  ndi_eval_opts opts;
  ndi_oob_info info;
  ndi_status st;

  for (i = 0; i < 10; ++i) q[i] = -1.0 + i * (4.0 / 9.0); /* Array::linspace(-1, 3, 10) */
  q[9] = 3.0;
  memset(&d, 0, sizeof d);
  d.dtype = NDI_F64;
  d.strategy = NDI_CUBIC_SPLINE;
  d.n = 3; d.lanes = 1; d.x_len = 3;
  d.x = x; d.data = y;
  d.memspace = NDI_MEM_HOST;
  d.validate = 1;
  d.left.kind = NDI_BC_NOT_A_KNOT; d.right.kind = NDI_BC_NOT_A_KNOT;
  st = ndi_interp1d_create(&d, &h);
  if (st != NDI_OK) { fprintf(stderr, "create: %d %s\n", (int)st, ndi_last_error_string()); return 1; }
  memset(&opts, 0, sizeof opts);   /* host queries, host output, default stream, AUTO formulation */
  memset(&info, 0, sizeof info);
  st = ndi_interp1d_eval(h, q, 10, out, 1, &opts, &info);
  if (st != NDI_OK) { fprintf(stderr, "eval: %d %s\n", (int)st, ndi_last_error_string()); return 2; }
  for (i = 0; i < 10; ++i) printf("%.17g\n", out[i]);
  {
    /* the same batch in chunks of 4 queries through a library-owned 2-slot device ring (ndi_interp1d_eval_ring:
     * interp_array for outputs that do not fit / need not stay in device memory) */
    ndi_ring_desc ring;
    unsigned long rows = 0;
    memset(&ring, 0, sizeof ring);
    ring.n_slots = 2;
    ring.chunk_queries = 4;
    opts.out_memspace = NDI_MEM_DEVICE;
    st = ndi_interp1d_eval_ring(h, q, 10, &ring, count_rows, &rows, &opts, &info);
    if (st != NDI_OK || rows != 10) { fprintf(stderr, "eval_ring: %d %s\n", (int)st, ndi_last_error_string()); return 3; }
  }
  {
    /* VectorExtensions::get_lower_index with the knots resident on the device */
    ndi_locator* loc = NULL;
    int64_t idx[10];
    st = ndi_locator_create(NDI_F64, 0, x, 3, NDI_MEM_HOST, &loc);
    if (st == NDI_OK) st = ndi_locator_eval(loc, q, 10, idx, NDI_MEM_HOST, NULL);
    ndi_locator_destroy(loc);
    if (st != NDI_OK || idx[0] != 0 || idx[3] != 1 || idx[9] != 1) {
      fprintf(stderr, "locator: %d %s\n", (int)st, ndi_last_error_string());
      return 4;
    }
  }
  ndi_interp1d_destroy(h);
  return 0;
}
