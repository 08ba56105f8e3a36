// Stand-alone probe of the persistent tile Cholesky's pieces (first hardware contact / debugging):
//   ct_probe <test> [N]   test 0: leaf core<4> inlined in a 256-thread kernel; 1: the same through the non-inlined call;
//                         2: chol_tiles_kernel on an N x N matrix (default 128); 6: the same with 8 waves; 7: the same with a lost tile (expects `abort 1` after the
//                         time-out, not a hang); prints max |L - L_ref|.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/ct_probe.hip -o tools/probes/bin/ct_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <thread>
#include <unistd.h>
#include "../../gumbi_amd/csrc/chol_tiles.hpp"
using namespace gmb;

__global__ __launch_bounds__(256, 2) void k_leaf_inline(LeafArgs a) {
  __shared__ __attribute__((aligned(16))) double lds[ct_lds_doubles(4)];
  potrf_leaf_core<4>(a, lds);
}
__device__ __noinline__ void leaf_call4(const LeafArgs a_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* logdet, ct_g_i32* info,
                                       ct_lds_double* l3) {
  LeafArgs a = a_in;
  a.A = (double*)A; a.dinv16 = (double*)dinv16; a.logdet = (double*)logdet; a.info = (int32_t*)info;
  potrf_leaf_core<4>(a, (double*)l3);
}
__global__ __launch_bounds__(256, 2) void k_leaf_call(LeafArgs a) {
  __shared__ __attribute__((aligned(16))) double lds[ct_lds_doubles(4)];
  leaf_call4(a, (ct_g_double*)a.A, (ct_g_double*)a.dinv16, (ct_g_double*)a.logdet, (ct_g_i32*)a.info, (ct_lds_double*)lds);
}
#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(s_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const int test = argc > 1 ? atoi(argv[1]) : 0;
  const int N = argc > 2 ? atoi(argv[2]) : 128;
  const int nct = (N + 127) / 128, nrt = (N + 1 + 127) / 128;
  const int64_t Np = 128LL * nct, Nr = 128LL * nrt, ld = Nr;
  std::vector<double> A((size_t)ld * Np, 0.0), R;
  srand(1);
  // SPD: A = B B^T / n + I on the real part, identity padding, row N = "y"
  std::vector<double> B((size_t)N * 8);
  for (auto& b : B) b = rand() / (double)RAND_MAX - 0.5;
  for (int c = 0; c < Np; ++c)
    for (int r = c; r < Nr; ++r) {
      double v = 0.0;
      if (r < N && c < N) { for (int k = 0; k < 8; ++k) v += B[(size_t)r * 8 + k] * B[(size_t)c * 8 + k]; if (r == c) v += 1.0; }
      else if (r == N && c < N) v = 0.01 * (c % 7);
      else if (r == c) v = 1.0;
      A[r + (size_t)c * ld] = v;
    }
  R = A;
  for (int c = 0; c < N; ++c) {  // reference (rows to Nr ride along)
    const double l = std::sqrt(R[c + (size_t)c * ld]);
    for (int r = c; r < Nr; ++r) R[r + (size_t)c * ld] /= (r == c ? 1.0 : l);
    R[c + (size_t)c * ld] = l;
    for (int c2 = c + 1; c2 < N; ++c2) {
      const double f = R[c2 + (size_t)c * ld];
      for (int r = c2; r < Nr; ++r) R[r + (size_t)c2 * ld] -= f * R[r + (size_t)c * ld];
    }
  }
  double *dA, *dinv, *dscal; int32_t* dinfo; uint32_t* dct; unsigned long long* dtr;
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dinv, (size_t)nct * 2048 * 8)); CK(hipMalloc(&dscal, 64 * 8));
  CK(hipMalloc(&dinfo, 4)); CK(hipMalloc(&dct, (4 + (size_t)nrt * nct + 3 * nct) * 4));
  const int ntasks = ct_task_count(nct, nrt);
  CK(hipMalloc(&dtr, (size_t)ntasks * 32));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dscal, 0, 64 * 8)); CK(hipMemset(dinfo, 0, 4)); CK(hipMemset(dct, 0, (4 + (size_t)nrt * nct + 3 * nct) * 4)); CK(hipMemset(dtr, 0, (size_t)ntasks * 32));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (test < 2 || test == 5) {
    LeafArgs a{}; a.A = dA; a.lda = ld; a.nvalid = N < 128 ? N : 128; a.dinv16 = dinv; a.logdet = dscal; a.info = dinfo; a.row0 = 0; a.dbg = nullptr;
    // timing: the same block factored 50 times from a fresh copy each (copy kernel + leaf, events around the leaves only would
    // need one pair each: the mean of (copy + leaf) minus the mean of copy alone is close enough for a comparison)
    double* dA0; CK(hipMalloc(&dA0, A.size() * 8)); CK(hipMemcpy(dA0, dA, A.size() * 8, hipMemcpyDeviceToDevice));
    double* ddbg; CK(hipMalloc(&ddbg, 64 * 8)); CK(hipMemset(ddbg, 0, 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.f;
    for (int rep = 0; rep < 52; ++rep) {
      CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
      a.dbg = rep == 51 ? ddbg : nullptr;
      CK(hipEventRecord(e0, st));
      if (test == 0) hipLaunchKernelGGL(k_leaf_inline, dim3(1), dim3(256), 0, st, a);
      else if (test == 1) hipLaunchKernelGGL(k_leaf_call, dim3(1), dim3(256), 0, st, a);
      else hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(512), 0, st, a);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && rep < 51) tot += ms;
    }
    double hd[64]; CK(hipMemcpy(hd, ddbg, 64 * 8, hipMemcpyDeviceToHost));
    printf("test %d: leaf launch %.2f us (events, mean of 49); stamps (us from entry): loaded %.2f", test, tot / 49 * 1e3, (hd[1] - hd[0]) * 0.01);
    for (int s2 = 0; s2 < 7; ++s2) printf(" | s%d solve %.2f bar %.2f upd %.2f", s2, (hd[8 + 3 * s2] - hd[0]) * 0.01, (hd[9 + 3 * s2] - hd[0]) * 0.01, (hd[10 + 3 * s2] - hd[0]) * 0.01);
    printf(" | loop end %.2f logdet %.2f end %.2f\n", (hd[2] - hd[0]) * 0.01, (hd[3] - hd[0]) * 0.01, (hd[4] - hd[0]) * 0.01);
    CK(hipMemset(dscal, 0, 64 * 8));
    CK(hipMemcpyAsync(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice, st));
    a.dbg = nullptr;
    if (test == 0) hipLaunchKernelGGL(k_leaf_inline, dim3(1), dim3(256), 0, st, a);
    else if (test == 1) hipLaunchKernelGGL(k_leaf_call, dim3(1), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(512), 0, st, a);
  } else {
    CholTilesArgs g{}; g.A = dA; g.ld = ld; g.nct = nct; g.nrt = nrt; g.N = N; g.dinv16 = dinv; g.logdet = dscal; g.info = dinfo;
    g.ctl = dct; g.flags = dct + 4; g.half = dct + 4 + (size_t)nrt * nct; g.prog = g.half + 2 * nct; g.ntasks = ntasks; g.timeout_us = 2000000u; g.dbg = dtr;
    const int grid = ntasks < 512 ? ntasks : 512;
    if (test == 2) hipLaunchKernelGGL(chol_tiles_kernel<4>, dim3(grid), dim3(256), 0, st, g);
    if (test == 6) hipLaunchKernelGGL(chol_tiles_kernel<8>, dim3(grid < 256 ? grid : 256), dim3(512), 0, st, g);
    if (test == 7) {  // a lost tile: ticket 0 is never handed out, every other task waits for tile (0, 0) -- the watchdog must end the launch
      const uint32_t one = 1;
      CK(hipMemcpy(dct, &one, 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(chol_tiles_kernel<8>, dim3(grid < 256 ? grid : 256), dim3(512), 0, st, g);
    }
  }
  CK(hipGetLastError());
  auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { printf("stream error: %s\n", hipGetErrorString(q)); return 3; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 8.0) { printf("test %d N %d: HANG (> 8 s)\n", test, N); fflush(stdout); _exit(4); }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  CK(hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost));
  uint32_t ctl[4]; CK(hipMemcpy(ctl, dct, 16, hipMemcpyDeviceToHost));
  double hs[2]; CK(hipMemcpy(hs, dscal, 16, hipMemcpyDeviceToHost));
  int32_t info; CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
  double err = 0.0, ld_ref = 0.0;
  const bool leaf_only = test < 2 || test == 5;
  const int cmax = leaf_only ? (N < 128 ? N : 128) : N, rmax = leaf_only ? 128 : (int)Nr;
  for (int c = 0; c < cmax; ++c) {
    ld_ref += std::log(R[c + (size_t)c * ld]);
    for (int r = c; r < rmax && r <= N; ++r) err = std::fmax(err, std::fabs(A[r + (size_t)c * ld] - R[r + (size_t)c * ld]));
  }
  printf("test %d N %d: done in %.3f s, tickets %u abort %u info %d, max |L - Lref| = %.3e, logdet %.12g (ref %.12g)\n", test, N, secs, ctl[0], ctl[1], info, err, hs[0], ld_ref);
  return 0;
}
