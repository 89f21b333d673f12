// GPU probe (rounds 5-6): the skinny kernels of csrc/general_mlp.hip alone, one launch each over three 1024 x 1024
// networks and 64 rows, timed with HIP events; built in variants (-DGS_WAVES=.., -DGS_KB=.., -DGS_PF=..):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iomnisafe_amd/csrc tools/skinny_probe.hip -o /tmp/probe
// Round 6: the forward / backward-data launches with the fused features switched on one by one (norm slots, outputs
// before the bias, partial outputs of the fused top layer, Gram problems, Gram norm in the backward epilogue).
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
  float *W, *X, *Y, *B, *A, *YL, *SL, *W2, *OS, *G;
  CK(hipMalloc(&W, (size_t)3 * H * H * 4));
  CK(hipMalloc(&X, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&Y, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&YL, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&A, (size_t)3 * R * ld * 4));
  CK(hipMalloc(&B, (size_t)3 * H * 4));
  CK(hipMalloc(&SL, (size_t)3 * (H / 16 + 1) * 16 * 4));
  CK(hipMalloc(&W2, (size_t)3 * 4 * H * 4));
  CK(hipMalloc(&OS, (size_t)3 * (H / 16 + 1) * 64 * 4 * 4));
  CK(hipMalloc(&G, (size_t)3 * 64 * 64 * 4));
  CK(hipMemset(W, 0, (size_t)3 * H * H * 4));
  CK(hipMemset(X, 0, (size_t)3 * R * ld * 4));
  CK(hipMemset(A, 0, (size_t)3 * R * ld * 4));
  CK(hipMemset(YL, 0, (size_t)3 * R * ld * 4));
  CK(hipMemset(B, 0, (size_t)3 * H * 4));
  CK(hipMemset(W2, 0, (size_t)3 * 4 * H * 4));
  CK(hipMemset(G, 0, (size_t)3 * 64 * 64 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // feature bits: 1 slot, 2 Ylin, 4 oslab, 8 Gram problems (fwd) / 16 Gram norm (bwd)
  const int variants[] = {0, 15, 0 | 32, 16 | 32};  // 64: Gram problems ALONE; 128: three small ordinary problems (N = 64) next to the real ones; 256: the Gram problems FIRST
  for (int feat : variants) {
    GSArgs g = {};
    g.R = R; g.fused = (feat & 31) != 0;
    const bool bwd = feat & 32;
    if (feat == 256)
      for (int i = 0; i < 3; ++i) {
        GSProb& q = g.p[g.nprob++];
        q.X = X + (size_t)i * R * ld; q.ldx = ld; q.W = q.X; q.ldw = ld; q.Y = G + (size_t)i * 4096; q.ldy = 64; q.N = R; q.K = H; q.act = -1;
      }
    for (int i = 0; i < 3; ++i) {
      if (feat == 64) {
        GSProb& q = g.p[g.nprob++];
        q.X = X + (size_t)i * R * ld; q.ldx = ld; q.W = q.X; q.ldw = ld; q.Y = G + (size_t)i * 4096; q.ldy = 64; q.N = R; q.K = H; q.act = -1;
        continue;
      }
      GSProb& p = g.p[g.nprob++];
      p.X = X + (size_t)i * R * ld; p.ldx = ld; p.W = W + (size_t)i * H * H; p.ldw = H; p.bias = B + i * H;
      p.Y = Y + (size_t)i * R * ld; p.ldy = ld; p.aux = A + (size_t)i * R * ld; p.ldaux = ld; p.N = H; p.K = H; p.act = 0;
      if (!bwd) {
        if (feat & 1) p.slot = SL + (size_t)i * (H / 16 + 1) * 4;
        if (feat & 2) p.Ylin = YL + (size_t)i * R * ld;
        if (feat & 4) { p.W2 = W2 + (size_t)i * 4 * H; p.ldw2 = H; p.N2 = 2; p.ldo2 = 4; p.oslab = OS + (size_t)i * (H / 16 + 1) * 256; }
        if (feat == 128) {
          GSProb& q = g.p[g.nprob++];
          q = p; q.Y = G + (size_t)i * 4096; q.ldy = 64; q.N = 64;
        }
        if (feat & 8) {
          GSProb& q = g.p[g.nprob++];
          q.X = p.X; q.ldx = ld; q.W = p.X; q.ldw = ld; q.Y = G + (size_t)i * 4096; q.ldy = 64; q.N = R; q.K = H; q.act = -1;
        }
      } else if (feat & 16) {
        p.nslot = SL + (size_t)i * (H / 16 + 1) * 4; p.G = G + (size_t)i * 4096; p.bvec = B + i * H; p.c2 = 0.002f;
        if (feat & 2) p.Zlin = YL + (size_t)i * R * ld;
      }
    }
    int nwg = 0;
    for (int y = 0; y < g.nprob; ++y) { g.tile0[y] = nwg; nwg += (g.p[y].ldy + 15) / 16; }
    for (int y = g.nprob; y <= GS_MAXPROB; ++y) g.tile0[y] = nwg;
    for (int it = 0; it < 2; ++it) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        if (!bwd) hipLaunchKernelGGL(gs_fwd_kernel<false>, dim3(nwg), dim3(64 * GS_WAVES), 0, 0, g);
        else hipLaunchKernelGGL(gs_bwd_kernel, dim3(nwg), dim3(64 * GS_WAVES), 0, 0, g);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("%s features=%d (1 slot 2 Ylin 4 oslab 8 Gram problems 16 Gram norm) H=%d R=%d : %.2f us per launch (back to back)\n",
           bwd ? "bwd" : "fwd", feat, H, R, ms * 1e3 / reps);
  }
  // ---- the top launch (loss + backward through a 2-wide top layer + norm slots), with phase stamps (-DGS_CLOCKS)
  {
    float *OUT, *DZ, *SP, *ACT, *LS, *DLS, *LP, *FIN, *ST, *LAM;
    int* STEP;
    CK(hipMalloc(&OUT, 3 * 64 * 4 * 4)); CK(hipMalloc(&DZ, 3 * 64 * 4 * 4)); CK(hipMalloc(&SP, 5 * 64 * 4));
    CK(hipMalloc(&ACT, 64 * 4 * 4)); CK(hipMalloc(&LS, 16)); CK(hipMalloc(&DLS, 64)); CK(hipMalloc(&LP, 3 * 16));
    CK(hipMalloc(&FIN, 3 * 8 * 4)); CK(hipMalloc(&ST, 64)); CK(hipMalloc(&LAM, 4)); CK(hipMalloc(&STEP, 12));
    CK(hipMemset(OUT, 0, 3 * 64 * 4 * 4)); CK(hipMemset(SP, 0, 5 * 64 * 4)); CK(hipMemset(ACT, 0, 64 * 4 * 4));
    CK(hipMemset(LS, 0, 16)); CK(hipMemset(LAM, 0, 4)); CK(hipMemset(STEP, 0, 12)); CK(hipMemset(OS, 0, (size_t)3 * (H / 16 + 1) * 64 * 4 * 4));
    for (int fused = 0; fused < 2; ++fused) {
      GSArgs g = {};
      g.R = R; g.fused = fused; g.nprob = 3;
      GLossArgs la = {};
      la.R = R; la.act_dim = 2; la.lda = 4; la.clip = 0.2f; la.loss_kind = 0; la.nets_mask = 7; la.nblk = 1;
      la.log_std = LS; la.lagrange = LAM; la.dls = DLS; la.lpart = LP; la.direct = 1; la.ld_act = 4; la.act = ACT;
      for (int q = 0; q < 5; ++q) la.sp[q] = SP + q * 64;
      la.scal = SP; la.actg = ACT;
      for (int i = 0; i < 3; ++i) {
        la.out[i] = OUT + i * 256; la.ldo[i] = 4; la.dz[i] = DZ + i * 256; la.ldz[i] = 4;
        GSProb& p = g.p[i];
        p.net = i; p.W = W2 + (size_t)i * 4 * H; p.ldw = H; p.N = i == 0 ? 2 : 1; p.K = H; p.act = 0;
        p.Y = Y + (size_t)i * R * ld; p.ldy = ld; p.aux = A + (size_t)i * R * ld; p.ldaux = ld;
        if (fused) {
          p.c2 = i ? 0.002f : 0.f; p.btop = B + i * H; p.tslot = SL + (size_t)i * (H / 16 + 1) * 4;
          p.nslot = SL + (size_t)(3 + i) * (H / 16 + 1) * 4; p.G = G + (size_t)i * 4096; p.bvec = B + i * H;
          p.Zlin = i ? YL + (size_t)i * R * ld : nullptr; p.oslab = OS + (size_t)i * (H / 16 + 1) * 256; p.ldo2 = 4; p.nbo = H / 16;
        }
      }
      GSTopFin tf = {};
      tf.stats = ST; tf.act_dim = 2;
      int nwg = 0;
      for (int y = 0; y < g.nprob; ++y) { g.tile0[y] = nwg; nwg += (g.p[y].ldy + 15) / 16; }
      for (int y = g.nprob; y <= GS_MAXPROB; ++y) g.tile0[y] = nwg;
      for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gs_top_kernel, dim3(nwg), dim3(256), 0, 0, g, la, tf);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("top fused=%d : %.2f us per launch (back to back)\n", fused, ms * 1e3 / reps);
#ifdef GS_CLOCKS
      long long clk[2][16];
      CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(gs_clk), sizeof(clk)));
#ifdef GS_CLOCKS_WAIT
      if (fused) {
        printf("  loads waited for group by group (wave 1 of tile 1), cycles:");
        for (int q = 1; q < 6; ++q) printf(" %lld", clk[0][8 + q] - clk[0][8 + q - 1]);
        printf("   (Gram 16 x 4 B, top weights 8 x 4 B, h / Zlin / biases, slabs 16 x 16 B, gathered operands)\n");
      }
#endif
      if (fused) {
        long long lc[16];
        CK(hipMemcpyFromSymbol(lc, HIP_SYMBOL(gs_lclk), sizeof(lc)));
        printf("  loss of tile 1, cycles since its entry:");
        for (int q = 1; q < 7; ++q) printf(" %lld", lc[q] - lc[0]);
        printf("   (1 log-prob + ratio, 2 surrogate, 3 gradient rows + log_std sums, 4 padding, 5 loss sums, 6 end)\n");
      }
      for (int t = 0; t < 2; ++t) {
        printf("  tile %d cycles since entry:", t);
        for (int q = 1; q < 8; ++q) printf(" %lld", clk[t][q] - clk[t][0]);
        printf("   (1 operands staged, 2 outputs summed, 3 loss, 4 writer's tail, 5 dZ tile, 6 norm chains, 7 slots written)\n");
      }
#endif
    }
  }
  return 0;
}
