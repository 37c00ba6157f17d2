// What a fork of the weight-gradient stream costs the main stream: N small dependent kernels back to back (a) alone, (b) with a
// hipEventRecord behind each, (c) launched through hipExtLaunchKernelGGL with the event as the kernel's own stop event (no marker
// packet), (d)/(e) the same two with a second stream waiting for the event and running a kernel.  The host is kept ahead by a
// long kernel in front of each run.   hipcc --offload-arch=gfx950 -O2 scripts/event_cost.hip -o /tmp/event_cost && /tmp/event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void tick(float* p) { p[threadIdx.x + blockIdx.x * blockDim.x] += 1.f; }
__global__ void hold(float* p, long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1.f;
}
__global__ void fill(int* p, long n, int v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void verify(const int* p, long n, int v, int* bad) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) if (p[i] != v) atomicAdd(bad, 1);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const unsigned flags = hipEventDisableTiming | (argc > 1 ? hipEventReleaseToDevice : 0);
    printf("events: %s\n", argc > 1 ? "release to device" : "default (release to system)");
    float *a, *b, *big; const long NB = 1L << 28;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&big, NB * 4));
    CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20)); CK(hipMemset(big, 0, NB * 4));
    hipStream_t sm, ss; CK(hipStreamCreate(&sm)); CK(hipStreamCreate(&ss));
    const int N = 1000;
    std::vector<hipEvent_t> ev(64);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const char* names[] = {"kernels only", "+ hipEventRecord", "+ stop event bound to the kernel (hipExtLaunchKernelGGL)",
                           "+ hipEventRecord + side stream waits and runs a kernel", "+ bound stop event + side stream waits and runs a kernel"};
    for (int mode = 0; mode < 5; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(hold, dim3(2048), dim3(256), 0, sm, big, NB, 40);       // ~15-20 ms
            CK(hipEventRecord(t0, sm));
            for (int i = 0; i < N; ++i) {
                hipEvent_t e = ev[i % 64];
                if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(tick, dim3(64), dim3(256), 0, sm, nullptr, e, 0, a);
                else hipLaunchKernelGGL(tick, dim3(64), dim3(256), 0, sm, a);
                if (mode == 1 || mode == 3) CK(hipEventRecord(e, sm));
                if (mode >= 3) { CK(hipStreamWaitEvent(ss, e, 0)); hipLaunchKernelGGL(tick, dim3(64), dim3(256), 0, ss, b); }
            }
            CK(hipEventRecord(t1, sm));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep) printf("%-62s %6.2f us per kernel on the main stream\n", names[mode], ms * 1e3f / N);
        }
    // ordering: a kernel of ~50 us fills 64 MB with the round number, bound stop event, the side stream waits and verifies
    {
        int *buf, *bad; const long n = 16L << 20; CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
        hipEvent_t back; CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));
        for (int r = 1; r <= 200; ++r) {
            hipExtLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, sm, nullptr, ev[r % 64], 0, buf, n, r);
            CK(hipStreamWaitEvent(ss, ev[r % 64], 0));
            hipExtLaunchKernelGGL(verify, dim3(1024), dim3(256), 0, ss, nullptr, back, 0, (const int*)buf, n, r, bad);
            CK(hipStreamWaitEvent(sm, back, 0));                                   // the next fill must not overtake the check
        }
        CK(hipDeviceSynchronize());
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("ordering through bound stop events: %d stale elements seen by the waiting stream (0 expected)\n", hb);
    }
    float h; CK(hipMemcpy(&h, a, 4, hipMemcpyDeviceToHost)); float h2; CK(hipMemcpy(&h2, b, 4, hipMemcpyDeviceToHost));
    printf("check: a[0] = %.0f (10000 expected), b[0] = %.0f (4000 expected)\n", h, h2);
    return 0;
}
