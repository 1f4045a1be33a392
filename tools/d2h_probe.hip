// d2h_probe.hip -- which engine moves a device-to-host copy of a reply group (8 MB)?  The wire path's group-wise reply
// downloads show up in kernel traces as __amd_rocclr_copyBuffer (a blit KERNEL: 256 workgroups x 512 threads that sit on
// every CU for the PCIe transfer's 150 us) instead of an SDMA transfer.  This probe issues the same copy in the forms
// the runtime distinguishes -- destination allocation flags, stream kind, a preceding cross-stream event wait, API entry
// point -- under rocprofv3 --kernel-trace --memory-copy-trace, so that the trace says which forms take which path.
//   hipcc --offload-arch=gfx950 -O2 tools/d2h_probe.hip -o tools/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void busy(double* p, int n) {
  double a = p[threadIdx.x];
  for (int i = 0; i < n; ++i) a = a * 1.0000001 + 0.5;
  p[threadIdx.x] = a;
}
__global__ void marker_kernel(int* p, int v) { if (threadIdx.x == 0) *p = v; }

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atol(argv[1]) : 8) << 20;
  void *dev, *dev2;
  CK(hipMalloc(&dev, bytes));
  CK(hipMalloc(&dev2, 4096));
  CK(hipMemset(dev, 1, bytes));
  struct { const char* name; unsigned flags; } allocs[] = {{"hipHostMallocDefault", hipHostMallocDefault},
                                                          {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                          {"hipHostMallocCoherent", hipHostMallocCoherent},
                                                          {"hipHostMallocPortable|Mapped", hipHostMallocPortable | hipHostMallocMapped}};
  hipStream_t nb, blocking, compute;
  CK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking));
  CK(hipStreamCreate(&blocking));
  CK(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
  hipEvent_t ev, t0, t1;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  int variant = 0;
  for (auto& al : allocs) {
    void* host;
    CK(hipHostMalloc(&host, bytes, al.flags));
    std::memset(host, 0, bytes);
    for (int form = 0; form < 5; ++form) {
      hipStream_t st = form == 1 ? blocking : nb;
      // a marker kernel whose argument encodes the variant: the trace is read against these
      hipLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, st, (int*)dev2, variant);
      if (form == 2 || form == 4) {   // the copy waits for an event of a compute stream (the product's shape)
        hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, compute, (double*)dev, 20000);
        CK(hipEventRecord(ev, compute));
        CK(hipStreamWaitEvent(st, ev, 0));
      }
      CK(hipEventRecord(t0, st));
      auto w0 = std::chrono::steady_clock::now();
      if (form == 3 || form == 4) CK(hipMemcpyDtoHAsync(host, (hipDeviceptr_t)dev, bytes, st));
      else CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
      auto w1 = std::chrono::steady_clock::now();
      CK(hipEventRecord(t1, st));
      CK(hipStreamSynchronize(st));
      CK(hipStreamSynchronize(compute));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      const char* forms[] = {"hipMemcpyAsync, non-blocking stream", "hipMemcpyAsync, blocking stream",
                             "hipMemcpyAsync, non-blocking stream, after an event of a compute stream",
                             "hipMemcpyDtoHAsync, non-blocking stream",
                             "hipMemcpyDtoHAsync, non-blocking stream, after an event of a compute stream"};
      std::printf("variant %2d  %-30s  %-80s  %.3f ms  %.1f GB/s  (call took %.1f us)\n", variant, al.name, forms[form], ms,
                  bytes / (ms * 1e6), std::chrono::duration<double, std::micro>(w1 - w0).count());
      ++variant;
    }
    CK(hipHostFree(host));
  }
  // the product's surroundings: many streams alive, uploads in flight on another stream, the copy issued by a second thread
  {
    void *host, *up_h, *up_d;
    CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&up_h, bytes, hipHostMallocDefault));
    CK(hipMalloc(&up_d, bytes));
    std::memset(host, 0, bytes);
    std::memset(up_h, 1, bytes);
    std::vector<hipStream_t> extra(24);
    for (auto& x : extra) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (auto& x : extra) hipLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, x, (int*)dev2, 99);   // every stream has run something
    CK(hipDeviceSynchronize());
    hipStream_t up;
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    for (int form = 0; form < 4; ++form) {
      hipLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, nb, (int*)dev2, variant);
      if (form >= 1) {   // an upload of the same size in flight
        for (int r = 0; r < 4; ++r) CK(hipMemcpyAsync(up_d, up_h, bytes, hipMemcpyHostToDevice, up));
      }
      if (form >= 2) {   // ... and compute on every other stream
        for (auto& x : extra) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, x, (double*)dev, 20000);
        CK(hipEventRecord(ev, extra[0]));
        CK(hipStreamWaitEvent(nb, ev, 0));
      }
      CK(hipEventRecord(t0, nb));
      if (form == 3) {
        std::thread th([&] { CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, nb)); });
        th.join();
      } else {
        CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, nb));
      }
      CK(hipEventRecord(t1, nb));
      CK(hipStreamSynchronize(nb));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      const char* forms[] = {"25 more streams alive", "+ four uploads of the same size in flight on another stream",
                             "+ kernels on every other stream, the copy behind an event of one", "+ issued by a second host thread"};
      std::printf("variant %2d  %-30s  %-80s  %.3f ms  %.1f GB/s\n", variant, "hipHostMallocDefault", forms[form], ms, bytes / (ms * 1e6));
      ++variant;
    }
  }
  // a registered malloc'd buffer
  {
    void* host = std::aligned_alloc(4096, bytes);
    std::memset(host, 0, bytes);
    CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    hipLaunchKernelGGL(marker_kernel, dim3(1), dim3(64), 0, nb, (int*)dev2, variant);
    CK(hipEventRecord(t0, nb));
    CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, nb));
    CK(hipEventRecord(t1, nb));
    CK(hipStreamSynchronize(nb));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, t0, t1));
    std::printf("variant %2d  %-30s  %-80s  %.3f ms  %.1f GB/s\n", variant, "hipHostRegister", "hipMemcpyAsync, non-blocking stream", ms,
                bytes / (ms * 1e6));
  }
  return 0;
}
