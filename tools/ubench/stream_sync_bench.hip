// What a cross-stream dependency costs the stream that waits (MI355X, ROCm 7.2): a chain of N ~20 us kernels on stream A with,
// between consecutive kernels, (0) nothing, (1) hipEventRecord on A, (2) hipEventRecord on A + hipStreamWaitEvent(B, ev) [the
// "post" side of an exchange], (3) hipStreamWaitEvent(A, evB) on an event of B that completed long ago [the "wait" side],
// (4) both (one exchange per kernel), (5) hipStreamWaitValue32 on a word that already holds the value,
// (6) hipStreamWriteValue32 on A (a signal another stream could poll).  Prints wall time per kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void spin(float* p, int iters) {
  float v = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}
__global__ void tiny(float* p) { p[0] = 1.f; }
int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  float* buf; CK(hipMalloc(&buf, 1 << 24));
  unsigned* word; CK(hipMalloc(&word, 4)); CK(hipMemset(word, 0, 4));
  hipEvent_t evA, evB; CK(hipEventCreateWithFlags(&evA, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evB, hipEventDisableTiming));
  const int N = 200, iters = 6000;
  CK(hipStreamWriteValue32(B, word, 7, 0));
  tiny<<<1, 64, 0, B>>>(buf + (1 << 20)); CK(hipEventRecord(evB, B)); CK(hipDeviceSynchronize());
  const char* names[] = {"plain chain", "eventRecord(A)", "eventRecord(A)+streamWait(B)", "streamWaitEvent(A, done event of B)",
                         "post + wait (one exchange per kernel, B runs a tiny kernel)", "streamWaitValue32(A, satisfied)", "streamWriteValue32(A)"};
  for (int mode = 0; mode < 7; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        spin<<<1024, 256, 0, A>>>(buf, iters);
        if (mode == 1) CK(hipEventRecord(evA, A));
        if (mode == 2) { CK(hipEventRecord(evA, A)); CK(hipStreamWaitEvent(B, evA, 0)); }
        if (mode == 3) CK(hipStreamWaitEvent(A, evB, 0));
        if (mode == 4) {
          CK(hipEventRecord(evA, A)); CK(hipStreamWaitEvent(B, evA, 0)); tiny<<<1, 64, 0, B>>>(buf + (1 << 20)); CK(hipEventRecord(evB, B));
          spin<<<1024, 256, 0, A>>>(buf, iters);
          CK(hipStreamWaitEvent(A, evB, 0));
        }
        if (mode == 5) CK(hipStreamWaitValue32(A, word, 7, hipStreamWaitValueGte, 0xffffffffu));
        if (mode == 6) CK(hipStreamWriteValue32(A, word, 7, 0));
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 1) printf("mode %d %-70s %8.2f us per %s\n", mode, names[mode], us / N, mode == 4 ? "pair of kernels" : "kernel");
    }
  }
  return 0;
}
