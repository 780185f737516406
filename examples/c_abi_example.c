/* examples/c_abi_example.c -- the C ABI from plain C99 (what a cgo / JNI / Rust FFI binding sees).
 * Builds a CubicSpline interpolator from host arrays, evaluates a batch into a host buffer and prints
 * the result; exits non-zero on any error.  Link: -lndinterp_hip (needs an MI355X at run time).
 *
 *   gcc -std=c99 -Wall -pedantic examples/c_abi_example.c -Iinclude -Lndarray-interp_amd -lndinterp_hip
 */
#include <stdio.h>
#include <string.h>

#include "ndinterp.h"

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
  ndi_interp1d_destroy(h);
  return 0;
}
