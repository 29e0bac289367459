// ext_event.hip -- what a lane fork costs the producing stream: a chain of N dependent ~30 us kernels on one stream,
//   mode 0  alone
//   mode 1  hipEventRecord between every two kernels (a barrier packet of its own in the queue)
//   mode 2  the same event attached to the kernel's OWN dispatch packet (hipExtLaunchKernelGGL stopEvent): no extra packet
//   mode 3 / 4  as 1 / 2, and a second stream waits for each event and runs a small kernel (the step's weight-gradient fork)
// build: hipcc -O2 --offload-arch=gfx950 ext_event.hip -o ext_event.bin      run: ./ext_event.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long ticks, int* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(out, 1);
}
__global__ void check_kernel(const int* produced, int expect, int* bad) {
    if (threadIdx.x == 0 && *produced < expect) atomicAdd(bad, 1);   // ran before its producer had finished
}

int main() {
    const int N = 200;
    hipStream_t mainS, sideS;
    CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sideS, hipStreamNonBlocking));
    hipEvent_t ev[N];
    for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    int *count, *bad;
    CK(hipMalloc(&count, 4)); CK(hipMalloc(&bad, 4));
    const unsigned long long ticks = 3000;                     // 30 us at 100 MHz
    for (int mode = 0; mode <= 4; ++mode) {
        double best = 1e30;
        int bad_h = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(count, 0, 4)); CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 2 || mode == 4) {
                    hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, mainS, nullptr, ev[i], 0, ticks, count);
                } else {
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, mainS, ticks, count);
                    if (mode == 1 || mode == 3) CK(hipEventRecord(ev[i], mainS));
                }
                if (mode >= 3) {
                    CK(hipStreamWaitEvent(sideS, ev[i], 0));
                    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, sideS, count, i + 1, bad);
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
            int b = 0;
            CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
            bad_h += b;
        }
        const char* names[] = {"chain alone", "hipEventRecord between kernels", "stop event on the kernel's own packet (hipExt)",
                               "hipEventRecord + side stream waits, runs a kernel", "hipExt stop event + side stream waits, runs a kernel"};
        printf("mode %d %-56s %7.2f us per kernel   side kernels that ran too early: %d\n", mode, names[mode], best / N, bad_h);
    }
    return 0;
}
