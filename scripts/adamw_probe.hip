// Round 5 probe: the optimizer sweep (30 B per parameter: master / gradient / both moments read, master / moments / bf16 shadow
// written) ran at 6.15 TB/s in round 4's trace and 5.4 TB/s in round 5's.  Variants of the same arithmetic on 182.4 M parameters:
//   0  the product kernel's shape: grid-stride, one 4-parameter group per thread and iteration, non-temporal accesses, 8192 blocks
//   1  the same with plain (temporal) accesses            2  two groups per thread and iteration (8 loads in flight)
//   3  2048 blocks                                         4  one contiguous range per block, 2048 blocks
//   5  four groups per thread and iteration                6  variant 2 with 1024 persistent blocks of 512 threads
// build: hipcc --offload-arch=gfx950 -O3 -w -o /tmp/adamw_probe scripts/adamw_probe.hip ; run: /tmp/adamw_probe
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ unsigned pack2(float a, float b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
    return (x >> 16) | (y & 0xffff0000u);
}
struct Hp { float lr, b1, b2, eps, wd, bc1, bc2s, gs; };
__device__ __forceinline__ void upd(f32x4& pp, const f32x4 gg, f32x4& mm, f32x4& vv, const Hp h) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gr = gg[e] * h.gs;
        pp[e] *= 1.0f - h.lr * h.wd;
        mm[e] = h.b1 * mm[e] + (1.0f - h.b1) * gr;
        vv[e] = h.b2 * vv[e] + (1.0f - h.b2) * gr * gr;
        pp[e] -= (h.lr / h.bc1) * mm[e] / (sqrtf(vv[e]) / h.bc2s + h.eps);
    }
}
template <bool NT, int U, bool RANGES>
__global__ void k_adamw(float* p, const float* g, float* m, float* v, u32x2* sh, long nv, Hp h) {
    long i, end, stride;
    if (RANGES) { const long per = (nv + gridDim.x - 1) / gridDim.x; i = (long)blockIdx.x * per + threadIdx.x; end = min(nv, (long)(blockIdx.x + 1) * per); stride = blockDim.x; }
    else { i = (long)blockIdx.x * blockDim.x + threadIdx.x; end = nv; stride = (long)gridDim.x * blockDim.x; }
    for (; i < end; i += stride * U) {
        f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < end) {
                if (NT) { pp[u] = __builtin_nontemporal_load((const f32x4*)p + j); gg[u] = __builtin_nontemporal_load((const f32x4*)g + j);
                          mm[u] = __builtin_nontemporal_load((const f32x4*)m + j); vv[u] = __builtin_nontemporal_load((const f32x4*)v + j); }
                else { pp[u] = ((const f32x4*)p)[j]; gg[u] = ((const f32x4*)g)[j]; mm[u] = ((const f32x4*)m)[j]; vv[u] = ((const f32x4*)v)[j]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < end) {
                upd(pp[u], gg[u], mm[u], vv[u], h);
                if (NT) { __builtin_nontemporal_store(pp[u], (f32x4*)p + j); __builtin_nontemporal_store(mm[u], (f32x4*)m + j); __builtin_nontemporal_store(vv[u], (f32x4*)v + j); }
                else { ((f32x4*)p)[j] = pp[u]; ((f32x4*)m)[j] = mm[u]; ((f32x4*)v)[j] = vv[u]; }
                sh[j] = u32x2{pack2(pp[u][0], pp[u][1]), pack2(pp[u][2], pp[u][3])};
            }
        }
    }
}
int main() {
    const long n = 182400000L, nv = n / 4;
    float *p, *g, *m, *v; u32x2* sh;
    const long pad = 8 << 20;
    hipMalloc(&p, n * 4 + pad); hipMalloc(&g, n * 4 + pad); hipMalloc(&m, n * 4 + pad); hipMalloc(&v, n * 4 + pad); hipMalloc(&sh, n * 2 + pad);
    printf("bases %p %p %p %p %p\n", p, g, m, v, sh);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    const Hp h{4e-5f, 0.9f, 0.999f, 1e-6f, 1e-4f, 0.1f, 0.0316f, 1.0f};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            if (ms < best) best = ms;
        }
        printf("%-64s %7.3f ms  %6.2f TB/s\n", name, best, 30.0 * n / best / 1e9);
    };
    // sweep: blocks x threads x groups per iteration x (non-temporal | plain)
    const int blocks[] = {128, 192, 256, 320, 384, 512, 768};
    const int threads[] = {64, 128, 192, 256, 384};
    for (int th : threads) for (int bl : blocks) {
        char name[96];
#define GO(NT_, U_) do { snprintf(name, sizeof name, "%5d x %4d, %d groups, %s", bl, th, U_, NT_ ? "non-temporal" : "plain"); \
        run(name, [&] { hipLaunchKernelGGL((k_adamw<NT_, U_, false>), dim3(bl), dim3(th), 0, 0, p, g, m, v, sh, nv, h); }); } while (0)
        GO(true, 1); GO(true, 2); GO(true, 3);
#undef GO
    }
    return 0;
}
