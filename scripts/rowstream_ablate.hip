// Diagnostic driver (round 5): times the row-streaming kernel stand-alone, without torch, so that timing-only ablation builds
// (-DRS_ABL_NOGELU / -DRS_ABL_NOSTORE / -DRS_ABL_NOMMA) can be compared: scripts/rowstream_ablate.sh builds and runs them.
// Shape: CASE 0 = stage-0 Mlp.fc1 forward (M 100352, K 96, N 384, bias + GELU + saved pre-activation);
//        CASE 1 = stage-0 fc2 dgrad (K 96 -> N 384, k-major weight, x gelu'(aux)); CASE 2 = stage-0 fc1 dgrad (384 -> 96)
#include "../medical-vision-langauge-transformer_amd/csrc/rowstream.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int cs = argc > 1 ? atoi(argv[1]) : 0;
    const int M = 100352, K = cs == 2 ? 384 : 96, N = cs == 2 ? 96 : 384;
    std::vector<unsigned short> h((size_t)M * 384);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 22) % 0x300) ^ ((i & 1) << 15);   // bf16 values ~ +-[0.008, 0.06]
    void *A, *W, *C, *P, *X; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&P, (size_t)M * N * 2);
    hipMalloc(&X, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(X, h.data(), (size_t)M * N * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
    GemmDev d{};
    d.M = M; d.N = N; d.K = K; d.A = A; d.lda = K; d.B = W; d.ldb = cs == 0 ? K : N; d.C = C; d.ldc = N;
    d.a_vec = d.b_vec = d.epi_vec = 1; d.split_k = 1;
    if (cs == 0) { d.epi = MVLT_EPI_BIAS | MVLT_EPI_GELU | MVLT_EPI_SAVE_PRE; d.bias = bias; d.pre = P; }
    if (cs == 1) { d.epi = MVLT_EPI_MUL_GELU_GRAD; d.aux = X; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (mvlt_rowstream_try(&d, cs != 0, nullptr) != 1) { printf("not taken\n"); return 1; }
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) mvlt_rowstream_try(&d, cs != 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        if (ms < best) best = ms;
    }
    printf("case %d: %.1f us\n", cs, best * 1e3);
    return 0;
}
