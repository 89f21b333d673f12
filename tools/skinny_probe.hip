// GPU probe (round 5): the skinny kernels of csrc/general_mlp.hip alone, one launch each over three 1024 x 1024
// networks and 64 rows, timed with HIP events; built in variants (-DGS_WAVES=.., -DGS_KB=.., -DGS_PF=..):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iomnisafe_amd/csrc tools/skinny_probe.hip -o /tmp/probe
#include <cstdio>
#include <vector>

#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../include/omnisafe_amd.h"
#include "../omnisafe_amd/csrc/skinny_mlp.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 1024, R = argc > 2 ? atoi(argv[2]) : 64, reps = 200;
  const int ld = H + (argc > 3 ? atoi(argv[3]) : 0);  // row padding of the activations (floats)
  float *W, *X, *Y, *B, *A;
  CK(hipMalloc(&W, (size_t)3 * H * H * 4));
  CK(hipMalloc(&X, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&Y, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&A, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&B, (size_t)3 * H * 4));
  CK(hipMemset(W, 0, (size_t)3 * H * H * 4));
  CK(hipMemset(X, 0, (size_t)3 * R * ld * 4));
  CK(hipMemset(A, 0, (size_t)3 * R * ld * 4));
  CK(hipMemset(B, 0, (size_t)3 * H * 4));
  GSArgs g = {};
  g.nprob = 3; g.R = R;
  for (int i = 0; i < 3; ++i) {
    GSProb& p = g.p[i];
    p.X = X + (size_t)i * R * ld; p.ldx = ld; p.W = W + (size_t)i * H * H; p.ldw = H; p.bias = B + i * H;
    p.Y = Y + (size_t)i * R * ld; p.ldy = ld; p.aux = A + (size_t)i * R * ld; p.ldaux = ld; p.N = H; p.K = H; p.act = 0;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int which = 0; which < 2; ++which) {
    for (int it = 0; it < 2; ++it) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        if (which == 0) hipLaunchKernelGGL(gs_fwd_kernel<false>, dim3((H + 15) / 16, 3), dim3(64 * GS_WAVES), 0, 0, g);
        else hipLaunchKernelGGL(gs_bwd_kernel, dim3((H + 15) / 16, 3), dim3(64 * GS_WAVES), 0, 0, g);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%s H=%d R=%d ld=%d waves=%d kb=%d pf=%d : %.2f us per launch (back to back)\n", which ? "bwd" : "fwd", H, R, ld,
           GS_WAVES, GS_KB, GS_PF, ms * 1e3 / reps);
  }
  return 0;
}
