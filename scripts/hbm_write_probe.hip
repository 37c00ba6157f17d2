// Round 5 probe: what does this chip sustain for WRITE-dominated streams?  The Swin stage-0 Mlp.fc1 product writes 154 MB
// and reads 19 MB; both the tile kernels and the row-streaming kernel stop at ~3.5 TB/s of those bytes.  Variants:
//   0  pure write, 16 B per lane, linear (1 KB contiguous per wave-instruction), grid-stride, 2048 blocks
//   1  pure write, 16 B per lane, the GEMM epilogue's shape: 16 rows x 64 B per wave-instruction (row stride 768 B), then the
//      next 64 B of the same rows
//   2  read 1 : write 8 (linear), the fc1 ratio
//   3  read 1 : write 1 (linear copy)                    4  pure read (sum kept alive)
//   5  pure write, 256 persistent blocks of 512 threads (one per CU), each writing ONE contiguous range (the persistent kernels' shape)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_write_probe scripts/hbm_write_probe.hip ; run: /tmp/hbm_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void k_write_linear(u32x4* out, long n16) {
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) out[i] = v;
}
__global__ void k_write_rows(u32x4* out, long rows) {          // rows of 768 B = 48 chunks of 16 B
    const int lane = threadIdx.x & 63, mr = lane & 15, g = lane >> 4;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (long rb = wave; rb * 16 < rows; rb += nwaves)
#pragma unroll
        for (int q = 0; q < 12; ++q) out[(rb * 16 + mr) * 48 + q * 4 + g] = v;
}
__global__ void k_rw(const u32x4* in, u32x4* out, long n16, int wr) {      // per 16-B chunk read, write `wr` chunks
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const u32x4 v = in[i];
        for (int w = 0; w < wr; ++w) out[i + (long)w * n16] = v;
    }
}
__global__ void k_read(const u32x4* in, long n16, unsigned* sink) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) { const u32x4 v = in[i]; acc ^= v[0] ^ v[3]; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write_ranges(u32x4* out, long n16) {
    const long per = (n16 + gridDim.x - 1) / gridDim.x, b = (long)blockIdx.x * per, e = b + per < n16 ? b + per : n16;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (long i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = v;
}

int main() {
    const long bytes = 154L << 20, n16 = bytes / 16;
    u32x4 *a, *b; unsigned* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, double moved, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            if (ms < best) best = ms;
        }
        printf("%-64s %7.1f us  %6.2f TB/s\n", name, best * 1e3, moved / best / 1e9);
    };
    run("0 pure write, linear 16 B/lane, 2048 blocks", bytes, [&] { hipLaunchKernelGGL(k_write_linear, dim3(2048), dim3(256), 0, 0, a, n16); });
    run("1 pure write, 16 rows x 64 B per instruction (epilogue shape)", bytes, [&] { hipLaunchKernelGGL(k_write_rows, dim3(2048), dim3(256), 0, 0, a, bytes / 768 / 16 * 16); });   // whole 16-row blocks only: stays inside the buffer
    run("2 read 1 : write 8 (fc1 ratio)", bytes * 9.0 / 8.0, [&] { hipLaunchKernelGGL(k_rw, dim3(2048), dim3(256), 0, 0, b, a, n16 / 8, 8); });
    run("3 read 1 : write 1 (copy), bytes = both directions", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_rw, dim3(2048), dim3(256), 0, 0, b, a, n16, 1); });
    run("4 pure read", bytes, [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, b, n16, sink); });
    run("5 pure write, 256 blocks x 512 threads, one contiguous range each", bytes, [&] { hipLaunchKernelGGL(k_write_ranges, dim3(256), dim3(512), 0, 0, a, n16); });
    run("6 pure write, linear, 8192 blocks", bytes, [&] { hipLaunchKernelGGL(k_write_linear, dim3(8192), dim3(256), 0, 0, a, n16); });
    return 0;
}
