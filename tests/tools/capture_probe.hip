// Minimal reproducer for the hipStreamEndCapture crash recorded in round 1 (gpurun_out/graph_err.txt): nested stream
// forks inside a capture, with per-call events that are destroyed while the capture is still open -- exactly what
// mimic_amd.trunk._WgradLane did under mimic_amd.mmvae._ModalityLanes through torch.cuda.Event / Stream.wait_event.
//   capture_probe <variant>
//     0  one level of forks (S0 -> S1 -> S0), events destroyed right after use              (what _ModalityLanes does)
//     1  nested forks (S0 -> S1 -> S2 -> S1 -> S0), events destroyed right after use        (the lanes under a capture)
//     2  nested forks, events kept alive until after EndCapture
//     3  nested forks from a second host thread (the autograd worker), events destroyed right after use
//     4  nested forks, inner stream S2 joined ONLY into S1 after S1 has already been joined into S0 (a late join)
//     5  nested forks, inner stream S2 never joined (expected: hipErrorStreamCaptureUnjoined, not a crash)
//     6  as 1, then what torch's CUDAGraph::capture_end does next: hipGraphGetNodes, hipGraphInstantiateWithFlags
//        (AutoFreeOnLaunch), hipGraphDestroy of the captured graph BEFORE the first launch
//     7  as 1 with streams from hipStreamCreateWithPriority (torch's stream pool)
//     8  6 + 7
//     9  as 8, capture begun under hipThreadExchangeStreamCaptureMode(relaxed) like torch's allocator calls
//    10  as 1, but S2 ENTERS the capture through an event of the ORIGIN stream S0 first (pre-fork), and only then waits
//        for S1's event: the nested dependency becomes an edge between two streams that are already capturing
// prints "variant N: <result>"; a crash shows as a signal exit status to the caller.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("variant %d: %s -> %s\n", variant, #x, hipGetErrorString(e_)); fflush(stdout); return 2; } } while (0)

__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }

static int variant = 0;
static std::vector<hipEvent_t> kept;

static hipError_t fork_join_event(hipStream_t from, hipStream_t to, bool keep) {
  hipEvent_t ev;
  hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) return e;
  if ((e = hipEventRecord(ev, from)) != hipSuccess) return e;
  if ((e = hipStreamWaitEvent(to, ev, 0)) != hipSuccess) return e;
  if (keep) kept.push_back(ev);
  else e = hipEventDestroy(ev);
  return e;
}

int main(int argc, char** argv) {
  variant = argc > 1 ? atoi(argv[1]) : 0;
  float* buf;
  CK(hipMalloc(&buf, 4 << 20));
  CK(hipMemset(buf, 0, 4 << 20));
  hipStream_t s0, s1, s2;
  const bool torch_like = variant >= 6, prio = variant >= 7;
  if (prio) {
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  } else {
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  }
  const bool keep = variant == 2;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s0, buf, 16384);
  CK(fork_join_event(s0, s1, keep));                       // fork the "modality" stream
  if (variant == 10) CK(fork_join_event(s0, s2, keep));    // pre-fork the lane from the origin as well
  auto body = [&]() -> int {
    hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s1, buf + 16384, 16384);
    if (variant >= 1) {
      if (variant == 9) { hipStreamCaptureMode m = hipStreamCaptureModeRelaxed; CK(hipThreadExchangeStreamCaptureMode(&m)); CK(hipThreadExchangeStreamCaptureMode(&m)); }
      CK(fork_join_event(s1, s2, keep));                   // fork the "lane" from the modality stream
      hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s2, buf + 32768, 16384);
      hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s1, buf + 49152, 16384);
      {   // what a framework's allocator asks while routing allocations to the capture's private pool
        hipStreamCaptureStatus st0, st1, st2;
        unsigned long long id0 = 0, id1 = 0, id2 = 0;
        CK(hipStreamGetCaptureInfo(s0, &st0, &id0));
        CK(hipStreamGetCaptureInfo(s1, &st1, &id1));
        CK(hipStreamGetCaptureInfo(s2, &st2, &id2));
        printf("variant %d: capture status/id  origin %d/%llu  fork %d/%llu  nested fork %d/%llu\n", variant, (int)st0, id0, (int)st1, id1, (int)st2, id2);
      }
      if (variant != 4 && variant != 5) CK(fork_join_event(s2, s1, keep));   // join the lane back
    }
    return 0;
  };
  int rc = 0;
  if (variant == 3) { std::thread t([&]() { rc = body(); }); t.join(); }
  else rc = body();
  if (rc) return rc;
  CK(fork_join_event(s1, s0, keep));                       // join the modality stream back
  if (variant == 4) { CK(fork_join_event(s2, s1, keep)); }  // late join of the lane into a stream that already joined
  hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, s0, buf, 16384);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(s0, &graph);
  printf("variant %d: hipStreamEndCapture -> %s, graph %p\n", variant, hipGetErrorString(e), (void*)graph);
  fflush(stdout);
  if (e == hipSuccess && graph) {
    hipGraphExec_t exec;
    if (torch_like) {
      size_t nn = 0;
      CK(hipGraphGetNodes(graph, nullptr, &nn));
      printf("variant %d: %zu nodes\n", variant, nn); fflush(stdout);
      CK(hipGraphInstantiateWithFlags(&exec, graph, hipGraphInstantiateFlagAutoFreeOnLaunch));
      printf("variant %d: instantiated\n", variant); fflush(stdout);
      CK(hipGraphDestroy(graph));
      printf("variant %d: captured graph destroyed\n", variant); fflush(stdout);
    } else
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, s0));
    CK(hipStreamSynchronize(s0));
    printf("variant %d: replay ok\n", variant);
  }
  return 0;
}
