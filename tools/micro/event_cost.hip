// What cross-stream synchronisation costs a stream whose kernels run back to back, and whether attaching the event to a kernel launch
// (hipExtLaunchKernelGGL start / stop events) is cheaper than a separate hipEventRecord packet.
// Stream A runs N kernels of ~20 us back to back (the host runs ahead); per iteration stream B runs one short kernel that must start
// behind A's kernel i (modes with an event).  Reported: wall time per iteration - the difference to `none` is what the mode costs A.
//   none        : no events at all (B's kernel unordered)
//   record      : hipEventRecord(e, A) behind A's kernel, hipStreamWaitEvent(B, e)                  (what libskf does today)
//   ext_stop    : A's kernel launched with hipExtLaunchKernelGGL(..., stopEvent = e), hipStreamWaitEvent(B, e)
//   ext_start   : the NEXT kernel of A launched with startEvent = e, then hipStreamWaitEvent(B, e)
//   wait        : A waits for an event B recorded one iteration ago (complete long before)
//   record+wait : both packets per iteration
//   hipcc --offload-arch=gfx950 -O2 tools/micro/event_cost.hip -o /tmp/event_cost && /tmp/event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream_add(float* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += v;
}

__global__ void set_word(int* w, int v) { *w = v; }
__global__ void copy_word(const int* w, int* out) { *out = *w; }

int main() {
  const size_t n = (size_t)1 << 24;                  // 64 MB read + write: ~25 us
  float *x, *y;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, 4096));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(y, 0, 4096));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  const int N = 400;
  std::vector<hipEvent_t> ev(2 * N);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const char* names[] = {"none", "record", "ext_stop", "ext_start", "wait", "record+wait"};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      hipEvent_t pending_start = nullptr;
      for (int i = 0; i < N; ++i) {
        hipEvent_t e = ev[i], e2 = ev[N + i];
        if (mode == 4 || mode == 5) {                  // B records, A waits (B's kernel of the previous iteration: long complete)
          CK(hipEventRecord(e2, B));
          CK(hipStreamWaitEvent(A, e2, 0));
        }
        if (mode == 2) {
          hipExtLaunchKernelGGL(stream_add, dim3(2048), dim3(256), 0, A, nullptr, e, 0, x, n, 1.f);
        } else if (mode == 3 && pending_start) {
          hipExtLaunchKernelGGL(stream_add, dim3(2048), dim3(256), 0, A, pending_start, nullptr, 0, x, n, 1.f);
          CK(hipStreamWaitEvent(B, pending_start, 0));
          hipLaunchKernelGGL(stream_add, dim3(1), dim3(256), 0, B, y, (size_t)256, 1.f);
        } else {
          hipLaunchKernelGGL(stream_add, dim3(2048), dim3(256), 0, A, x, n, 1.f);
        }
        if (mode == 1 || mode == 5) CK(hipEventRecord(e, A));
        if (mode == 1 || mode == 2 || mode == 5) CK(hipStreamWaitEvent(B, e, 0));
        if (mode == 3) pending_start = e;              // "everything of A up to here is complete" = the start of A's next kernel
        else hipLaunchKernelGGL(stream_add, dim3(1), dim3(256), 0, B, y, (size_t)256, 1.f);
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      printf("%-12s %7.2f us per iteration\n", names[mode], us);
    }
  // correctness of the launch-attached stop event: B's kernel i must see A's kernel i complete.  A's kernel i sets a word to i (after
  // a long streaming pass, so that an unordered B would run first), B's kernel i copies the word into slot i.
  int* word; int* seen;
  CK(hipMalloc(&word, 4)); CK(hipMalloc(&seen, N * 4));
  CK(hipMemset(word, 0xff, 4)); CK(hipMemset(seen, 0xff, N * 4));
  CK(hipDeviceSynchronize());
  for (int i = 0; i < N; ++i) {
    hipLaunchKernelGGL(stream_add, dim3(2048), dim3(256), 0, A, x, n, 1.f);
    hipExtLaunchKernelGGL(set_word, dim3(1), dim3(1), 0, A, nullptr, ev[i], 0, word, i);
    CK(hipStreamWaitEvent(B, ev[i], 0));
    hipLaunchKernelGGL(copy_word, dim3(1), dim3(1), 0, B, word, seen + i);
  }
  CK(hipDeviceSynchronize());
  std::vector<int> h(N);
  CK(hipMemcpy(h.data(), seen, N * 4, hipMemcpyDeviceToHost));
  int early = 0;
  for (int i = 0; i < N; ++i) early += h[i] < i;
  printf("stop-event ordering: %d of %d consumer kernels ran before their producer had finished (must be 0)\n", early, N);
  return early != 0;
}
