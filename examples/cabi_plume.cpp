// The C ABI on its own: a C++ host with nothing but the HIP runtime and include/fluidnet_hip.h -- no torch, no Python -- runs
// the 128 x 128 plume (configs[0]: Jacobi-28) for `steps` time steps through fnx_simulate_step and prints FNV-1a hashes of
// the resulting fields.  tests/test_abi.py builds and runs it on the GPU box and compares the hashes with the Python path.
//   hipcc --offload-arch=gfx950 -O2 -I include examples/cabi_plume.cpp -L fluidnet_cxx_amd -lfluidnet_hip \
//         -Wl,-rpath,$PWD/fluidnet_cxx_amd -o examples/cabi_plume.bin && examples/cabi_plume.bin 20
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "fluidnet_hip.h"

#define CHECK_FNX(expr)                                                         \
  do {                                                                          \
    if ((expr) != FNX_OK) { fprintf(stderr, "%s: %s\n", #expr, fnx_last_error()); return 1; } \
  } while (0)
#define CHECK_HIP(expr)                                                         \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 1; } \
  } while (0)

static uint64_t fnv1a(const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 20;
  const int res = argc > 2 ? atoi(argv[2]) : 128;
  if (!fnx_device_name()) { fprintf(stderr, "no HIP device: %s\n", fnx_last_error()); return 2; }
  FnxGrid g = {};
  g.B = 1; g.D = 1; g.H = res; g.W = res; g.is3D = 0;
  const size_t n = (size_t)res * res;
  // createPlumeBCs (init_conditions.py:4-83): inlet on rows 0..3, |x - W/2| <= floor(W * 0.145): U = (0, 2), density 0.1
  std::vector<float> UBC(2 * n, 0.f), UBCm(2 * n, 1.f), rBC(n, 0.f), rBCm(n, 1.f);
  const int rad = (int)floor(res * 0.145), cx = res / 2;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < res; ++i) {
      const bool in = (i - cx) * (i - cx) <= rad * rad;
      UBCm[j * res + i] = 0.f; UBCm[n + j * res + i] = 0.f;
      if (in) { UBC[n + j * res + i] = 2.f; rBC[j * res + i] = 0.1f; rBCm[j * res + i] = 0.f; }
    }
  float *p, *U, *rho, *flags, *dUBC, *dUBCm, *drBC, *drBCm;
  CHECK_HIP(hipMalloc(&p, n * 4)); CHECK_HIP(hipMalloc(&U, 2 * n * 4)); CHECK_HIP(hipMalloc(&rho, n * 4)); CHECK_HIP(hipMalloc(&flags, n * 4));
  CHECK_HIP(hipMalloc(&dUBC, 2 * n * 4)); CHECK_HIP(hipMalloc(&dUBCm, 2 * n * 4)); CHECK_HIP(hipMalloc(&drBC, n * 4)); CHECK_HIP(hipMalloc(&drBCm, n * 4));
  CHECK_HIP(hipMemset(p, 0, n * 4)); CHECK_HIP(hipMemset(U, 0, 2 * n * 4)); CHECK_HIP(hipMemset(rho, 0, n * 4));
  CHECK_HIP(hipMemcpy(dUBC, UBC.data(), 2 * n * 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dUBCm, UBCm.data(), 2 * n * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(drBC, rBC.data(), n * 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(drBCm, rBCm.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));
  CHECK_FNX(fnx_empty_domain(&g, flags, 1, s));
  const size_t ws_bytes = fnx_workspace_bytes(&g, FNX_OP_STEP);
  void* ws;
  CHECK_HIP(hipMalloc(&ws, ws_bytes));
  FnxStepParams prm = {};
  prm.dt = 0.1f; prm.maccormack_strength = 0.6f; prm.sample_outside_fluid = 0; prm.buoyancy_scale = 0.25f;
  prm.gravity_vec[0] = 0.f; prm.gravity_vec[1] = -1.f; prm.gravity_vec[2] = 0.f;
  prm.operating_density = 0.f; prm.p_tol = 0.f; prm.jacobi_iter = 28; prm.method = 0;
  FnxState st = {};
  st.p = p; st.U = U; st.density = rho; st.flags = flags; st.UBC = dUBC; st.UBCInvMask = dUBCm; st.densityBC = drBC; st.densityBCInvMask = drBCm;
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0)); CHECK_HIP(hipEventCreate(&e1));
  CHECK_HIP(hipEventRecord(e0, s));
  for (int it = 0; it < steps; ++it) {
    prm.static_flags = it == 0 ? 0 : (it == 1 ? 3 : 7);          // flags and BC arrays never change here
    CHECK_FNX(fnx_simulate_step(&g, &prm, &st, ws, ws_bytes, s));
  }
  CHECK_HIP(hipEventRecord(e1, s));
  CHECK_HIP(hipStreamSynchronize(s));
  float ms = 0.f;
  CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> h(2 * n);
  CHECK_HIP(hipMemcpy(h.data(), U, 2 * n * 4, hipMemcpyDeviceToHost));
  printf("U %016llx\n", (unsigned long long)fnv1a(h.data(), 2 * n * 4));
  CHECK_HIP(hipMemcpy(h.data(), rho, n * 4, hipMemcpyDeviceToHost));
  printf("density %016llx\n", (unsigned long long)fnv1a(h.data(), n * 4));
  CHECK_HIP(hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost));
  printf("p %016llx\n", (unsigned long long)fnv1a(h.data(), n * 4));
  printf("%d steps of the %dx%d plume on %s: %.1f us per step\n", steps, res, res, fnx_device_name(), ms / steps * 1e3);
  return 0;
}
