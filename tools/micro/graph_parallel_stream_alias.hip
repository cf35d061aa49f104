// Attempted reproducer for the SIGSEGV below hipGraphLaunch of a MULTI-BRANCH graph (profiles/r05y_two_stream_graph_crash.txt), without libskf.
//
// Disassembly of hip::Graph::UpdateStreams(hip::Stream* launch, const std::vector<hip::Stream*>& parallel) in the libamdhip64.so that
// ships with torch 2.10+rocm7.0 (function at .text+0xaed90, fault at +0xb1 = the load of parallel[i]->field_0x1a8):
//     streams_.resize(max_streams_);  streams_[0] = launch;
//     for (i = 0, k = 1; k < streams_.size(); ++i)                     // <- i is NOT bounded by parallel.size()
//       if (queue_of(parallel[i]) != queue_of(launch)) streams_[k++] = parallel[i];
// An internal stream of the graph exec that compares equal to the launch stream (same underlying queue object) is skipped, and the
// loop then reads past the end of the exec's stream vector.  Hypothesis tested here: streams are dealt over a small pool of hardware
// queues in creation order, so the number of streams a process created before decides whether the exec's stream aliases the launch
// stream.  RESULT (profiles/r06b_graph_alias.txt, both HIP runtimes of the image): none of the 72 cases below faults - the hypothesis
// about the trigger is not confirmed; the unbounded loop is in the disassembly either way, and the crash of
// profiles/r05y_two_stream_graph_crash.txt has only ever been seen in the pytest process that ran two test files.
//
// Each case runs in a forked child (the parent never touches HIP): `pre` dummy streams are created and used, then a two-branch graph
// is captured on a launch stream (fork to a side stream through an event, join back), instantiated and launched.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/graph_parallel_stream_alias.hip -o /tmp/graph_alias && /tmp/graph_alias
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

__global__ void touch(float* p, float v) { p[threadIdx.x] += v; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); _exit(3); } } while (0)

static int child(int pre, int null_launch, int destroy_pre) {
  float *a, *b;
  CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256));
  CK(hipMemset(a, 0, 256)); CK(hipMemset(b, 0, 256));
  std::vector<hipStream_t> dummies(pre);
  for (auto& s : dummies) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); hipLaunchKernelGGL(touch, 1, 64, 0, s, a, 0.f); }
  CK(hipDeviceSynchronize());
  if (destroy_pre) for (auto& s : dummies) CK(hipStreamDestroy(s));
  hipStream_t launch = nullptr, side;
  if (!null_launch) CK(hipStreamCreateWithFlags(&launch, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t fork_e, join_e;
  CK(hipEventCreateWithFlags(&fork_e, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join_e, hipEventDisableTiming));
  hipStream_t cap = launch;
  if (null_launch) CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));   // capture needs a real stream; the LAUNCH is on stream 0
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork_e, cap)); CK(hipStreamWaitEvent(side, fork_e, 0));
  for (int i = 0; i < 4; ++i) { hipLaunchKernelGGL(touch, 1, 64, 0, cap, a, 1.f); hipLaunchKernelGGL(touch, 1, 64, 0, side, b, 1.f); }
  CK(hipEventRecord(join_e, side)); CK(hipStreamWaitEvent(cap, join_e, 0));
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex, launch));
  CK(hipStreamSynchronize(launch));
  float h = 0.f;
  CK(hipMemcpy(&h, a, 4, hipMemcpyDeviceToHost));
  return h == 12.f ? 0 : 4;
}

// Second experiment: what the test process did - many build / replay / destroy cycles in ONE process, the destroy order of
// skf_model_destroy (exec, events, side stream), every `leak`-th cycle leaving its streams and exec alive, launches on stream 0
// (torch's current stream) or on one long-lived stream.
static int cycles(int n, int leak, int null_launch, int width) {
  float* a;
  CK(hipMalloc(&a, 4096)); CK(hipMemset(a, 0, 4096));
  hipStream_t launch = nullptr, cap;
  if (!null_launch) CK(hipStreamCreateWithFlags(&launch, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  for (int c = 0; c < n; ++c) {
    std::vector<hipStream_t> side(width);
    std::vector<hipEvent_t> ev(2 * width);
    for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipStream_t cs = null_launch ? cap : launch;
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    for (int w = 0; w < width; ++w) { CK(hipEventRecord(ev[2 * w], cs)); CK(hipStreamWaitEvent(side[w], ev[2 * w], 0)); }
    for (int i = 0; i < 3; ++i) {
      hipLaunchKernelGGL(touch, 1, 64, 0, cs, a, 1.f);
      for (int w = 0; w < width; ++w) hipLaunchKernelGGL(touch, 1, 64, 0, side[w], a + 64 * (w + 1), 1.f);
    }
    for (int w = 0; w < width; ++w) { CK(hipEventRecord(ev[2 * w + 1], side[w])); CK(hipStreamWaitEvent(cs, ev[2 * w + 1], 0)); }
    CK(hipStreamEndCapture(cs, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ex, launch));
    if (leak && c % leak == leak - 1) continue;              // (a model the garbage collector has not reached yet)
    CK(hipGraphExecDestroy(ex));                              // no synchronisation in front: skf_model_destroy has none either
    for (auto& e : ev) CK(hipEventDestroy(e));
    for (auto& s : side) CK(hipStreamDestroy(s));
  }
  CK(hipDeviceSynchronize());
  return 0;
}

int main() {
  int crashes = 0;
  for (int null_launch = 0; null_launch < 2; ++null_launch)
    for (int width = 1; width <= 3; ++width)
      for (int leak = 0; leak <= 3; ++leak) {
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) _exit(cycles(60, leak, null_launch, width));
        int st = 0;
        waitpid(pid, &st, 0);
        const bool sig = WIFSIGNALED(st);
        crashes += sig;
        printf("60 build / replay / destroy cycles, %d side stream(s), every %d-th cycle leaked, launch on %s: %s %d\n", width, leak,
               null_launch ? "stream 0" : "a long-lived stream", sig ? "SIGNAL" : "exit", sig ? WTERMSIG(st) : WEXITSTATUS(st));
      }
  printf("%d of 24 cycle cases died on a signal\n", crashes);
  crashes = 0;
  for (int null_launch = 0; null_launch < 2; ++null_launch)
    for (int destroy_pre = 0; destroy_pre < 2; ++destroy_pre)
      for (int pre = 0; pre < 12; ++pre) {
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0) _exit(child(pre, null_launch, destroy_pre));
        int st = 0;
        waitpid(pid, &st, 0);
        const bool sig = WIFSIGNALED(st);
        crashes += sig;
        printf("launch on %s, %2d streams created%s before: %s %d\n", null_launch ? "stream 0   " : "a new stream", pre,
               destroy_pre ? " and destroyed" : "              ", sig ? "SIGNAL" : "exit", sig ? WTERMSIG(st) : WEXITSTATUS(st));
      }
  printf("%d of 48 cases died on a signal\n", crashes);
  return 0;
}
