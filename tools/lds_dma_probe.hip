// tools/lds_dma_probe.hip -- round 6: semantics of global_load_lds_dwordx4 on gfx950 (per-lane global address, wave-uniform
// LDS base in M0): where do a lane's 16 bytes land?  Each lane loads the 16-byte record `perm[lane]` of a table of 64 records
// {id, id + 0.5}; the LDS region is then dumped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ void k(const char* g, const unsigned* offs, dbl2* out) {
  __shared__ __attribute__((aligned(16))) char s[2048];
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned i = threadIdx.x; i < 128; i += 64) reinterpret_cast<dbl2*>(s)[i] = dbl2{-1.0, -1.0};
  __syncthreads();
  const unsigned o = offs[lane];
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + o), (__attribute__((address_space(3))) void*)s, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[lane] = reinterpret_cast<const dbl2*>(s)[lane];
  out[64 + lane] = reinterpret_cast<const dbl2*>(s)[64 + lane];
}
int main() {
  std::vector<double> tab(128);
  for (int i = 0; i < 64; ++i) { tab[2 * i] = i; tab[2 * i + 1] = i + 0.5; }
  std::vector<unsigned> offs(64);
  for (int i = 0; i < 64; ++i) offs[i] = ((i * 37 + 5) % 64) * 16;
  char* g; unsigned* d_o; dbl2* d_out;
  hipMalloc(&g, 1024); hipMalloc(&d_o, 256); hipMalloc(&d_out, 2048);
  hipMemcpy(g, tab.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(d_o, offs.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, d_o, d_out);
  std::vector<double> out(256);
  hipMemcpy(out.data(), d_out, 2048, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; ++i) {
    const int id = (i * 37 + 5) % 64;
    if (out[2 * i] != id || out[2 * i + 1] != id + 0.5) ok = 0;
  }
  printf("lane*16 layout %s; first slots:", ok ? "CONFIRMED" : "NOT confirmed");
  for (int i = 0; i < 12; ++i) printf(" %.1f", out[i]);
  printf(" ... tail untouched: %.1f %.1f\n", out[128], out[255]);
  return ok ? 0 : 1;
}
