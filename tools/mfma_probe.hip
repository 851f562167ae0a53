// mfma_probe.hip -- how v_mfma_f64_4x4x4f64 lays out its operands over the lanes and in which order / with which
// roundings it adds the four k-steps (tools/, exploration): is D[i][*] the SEQUENTIAL sum ((C + a0) + a1) + a2) + a3?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe && tools/bin/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void k_mfma(const double *a, const double *b, const double *c, double *d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

int main() {
  double *da, *db, *dc, *dd;
  hipMalloc(&da, 512), hipMalloc(&db, 512), hipMalloc(&dc, 512), hipMalloc(&dd, 512);
  double a[64], b[64], c[64], d[64];
  auto run = [&]() {
    hipMemcpy(da, a, 512, hipMemcpyHostToDevice);
    hipMemcpy(db, b, 512, hipMemcpyHostToDevice);
    hipMemcpy(dc, c, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(d, dd, 512, hipMemcpyDeviceToHost);
  };
  // 1. which D lanes does A lane L feed (B = 1 everywhere)?
  std::printf("A lane -> D lanes it feeds (B = 1, C = 0)\n");
  for (int L = 0; L < 64; L += 1) {
    for (int i = 0; i < 64; ++i) a[i] = i == L, b[i] = 1.0, c[i] = 0.0;
    run();
    std::printf("A %2d ->", L);
    for (int i = 0; i < 64; ++i)
      if (d[i] != 0.0) std::printf(" %d", i);
    std::printf("\n");
  }
  // 2. k classes: A lane 0..15 against B lane 0..15 (block 0?): D nonzero iff same k
  std::printf("A lane x B lane -> any D nonzero (same k and block)\n");
  for (int La = 0; La < 16; ++La) {
    std::printf("A %2d:", La);
    for (int Lb = 0; Lb < 64; ++Lb) {
      for (int i = 0; i < 64; ++i) a[i] = i == La, b[i] = i == Lb, c[i] = 0.0;
      run();
      bool any = false;
      for (int i = 0; i < 64; ++i) any |= d[i] != 0.0;
      if (any) std::printf(" %d", Lb);
    }
    std::printf("\n");
  }
  // 3. order and roundings: a = {1, e, e, -1} with e = 2^-53 in every permutation of k; sequential from C = 0:
  //    ((0 + 1) + e) + e) - 1 = 0 (each e is lost), while any pairing (e + e) first gives 2^-52
  std::printf("order / roundings (layout: A[i = l %% 4][k = l / 16], D[i = l / 16][j = l %% 4], block (l / 4) %% 4)\n");
  const double e = 0x1p-53;
  const double pats[][5] = {{1, e, e, -1, 0}, {e, e, 1, -1, 0}, {1, -1, e, e, 0}, {e, 1, e, -1, 0}, {1, e, -1, e, 0}, {0x1p53, 1, 1, -0x1p53, 0},
                            {1, e, e, e, 0}, {e, e, e, 1, 0}, {0.1, 0.2, 0.3, 0.4, 0}, {0.1, 0.2, 0.3, 0.4, 1e16}};
  for (const auto &pt : pats) {
    for (int l = 0; l < 64; ++l) a[l] = pt[l / 16], b[l] = 1.0, c[l] = pt[4];
    run();
    volatile double seq = pt[4];
    for (int k = 0; k < 4; ++k) seq = seq + pt[k];
    volatile double rev = pt[4];
    for (int k = 3; k >= 0; --k) rev = rev + pt[k];
    volatile double prod_first = 0.0;  // the four products summed first (sequentially), C added last
    for (int k = 0; k < 4; ++k) prod_first = prod_first + pt[k];
    prod_first = prod_first + pt[4];
    const long double ex = (long double)pt[0] + pt[1] + pt[2] + pt[3] + pt[4];
    std::printf("a = {%g, %g, %g, %g}, C = %g: mfma %a | sequential from C %a | reverse %a | products first, then C %a | exact ~ %La\n",
                pt[0], pt[1], pt[2], pt[3], pt[4], d[0], (double)seq, (double)rev, (double)prod_first, ex);
  }
  return 0;
}
