// Torch-free use of libmmrag.so: the C-ABI of include/mmrag.h driven from plain HIP/C++.
//
//   hipcc -O2 --offload-arch=gfx950 -I include examples/c_abi_search.cpp -L multimodal_rag_amd/lib -lmmrag \
//         -Wl,-rpath,$PWD/multimodal_rag_amd/lib -o /tmp/c_abi_search && /tmp/c_abi_search [rows] [batch]
//
// Builds a unit-norm fp16 corpus on the device (through mmrag_append_rows, the reference's collection.add site,
// embedder.py:514-523), plants each query as a noisy copy of a known row, runs mmrag_cosine_topk
// (collection.query, embedder.py:595-601) and checks that the planted row comes back first.  Prints the
// timing and the achieved HBM rate; exit code 0 = all planted rows found.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mmrag.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)
#define MM_OK(x)                                                                   \
    do {                                                                           \
        if ((x) != MMRAG_OK) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, mmrag_last_error());                   \
            return 3;                                                              \
        }                                                                          \
    } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static float gauss() {  // xorshift + Box-Muller, enough for a demo corpus
    auto u = []() {
        rng_state ^= rng_state << 13;
        rng_state ^= rng_state >> 7;
        rng_state ^= rng_state << 17;
        return (float)((rng_state >> 11) + 1) / 9007199254740993.0f;
    };
    return sqrtf(-2.0f * logf(u())) * cosf(6.2831853f * u());
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 200000;
    const int B = argc > 2 ? atoi(argv[2]) : 64;
    const int d = 768, k = 5, dtype = MMRAG_F16;
    const long long ld = mmrag_padded_dim(d, dtype);
    printf("libmmrag ABI %d: corpus %lld x %d fp16 (row stride %lld), %d queries, top-%d\n", mmrag_abi_version(), n, d, ld, B, k);

    void *corpus = nullptr, *q = nullptr, *ws = nullptr;
    float *stage = nullptr, *scores = nullptr;
    int64_t *rows = nullptr;
    HIP_OK(hipMalloc(&corpus, (size_t)n * ld * 2));
    HIP_OK(hipMemset(corpus, 0, (size_t)n * ld * 2));
    HIP_OK(hipMalloc(&q, (size_t)B * ld * 2));
    HIP_OK(hipMemset(q, 0, (size_t)B * ld * 2));
    const long long chunk = 16384;
    HIP_OK(hipMalloc(&stage, (size_t)chunk * d * 4));
    std::vector<float> host((size_t)chunk * d), qhost((size_t)B * d);
    std::vector<long long> planted(B);
    for (int b = 0; b < B; ++b) planted[b] = (n / B) * b + (n / B) / 2;
    for (long long lo = 0; lo < n; lo += chunk) {
        const long long m = n - lo < chunk ? n - lo : chunk;
        for (long long r = 0; r < m; ++r) {
            float nrm = 0.f;
            float *row = &host[(size_t)r * d];
            for (int c = 0; c < d; ++c) {
                row[c] = gauss();
                nrm += row[c] * row[c];
            }
            nrm = 1.0f / sqrtf(nrm);
            for (int c = 0; c < d; ++c) row[c] *= nrm;
            for (int b = 0; b < B; ++b)
                if (planted[b] == lo + r) {  // query b = this row + a little noise, renormalised
                    float qn = 0.f;
                    float *qq = &qhost[(size_t)b * d];
                    for (int c = 0; c < d; ++c) {
                        qq[c] = row[c] + 0.002f * gauss();
                        qn += qq[c] * qq[c];
                    }
                    qn = 1.0f / sqrtf(qn);
                    for (int c = 0; c < d; ++c) qq[c] *= qn;
                }
        }
        HIP_OK(hipMemcpy(stage, host.data(), (size_t)m * d * 4, hipMemcpyHostToDevice));
        MM_OK(mmrag_append_rows(corpus, n, ld, dtype, lo, stage, m, d, nullptr));
    }
    HIP_OK(hipMemcpy(stage, qhost.data(), (size_t)B * d * 4, hipMemcpyHostToDevice));
    MM_OK(mmrag_append_rows(q, B, ld, dtype, 0, stage, B, d, nullptr));

    const size_t ws_bytes = mmrag_cosine_topk_workspace_bytes(B, n, k);
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(&scores, (size_t)B * k * 4));
    HIP_OK(hipMalloc(&rows, (size_t)B * k * 8));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        MM_OK(mmrag_cosine_topk(q, corpus, B, n, d, ld, dtype, k, 0, nullptr, scores, rows, ws, ws_bytes, nullptr));
    const int iters = 20;
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i)
        MM_OK(mmrag_cosine_topk(q, corpus, B, n, d, ld, dtype, k, 0, nullptr, scores, rows, ws, ws_bytes, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;

    std::vector<int64_t> hr((size_t)B * k);
    std::vector<float> hs((size_t)B * k);
    HIP_OK(hipMemcpy(hr.data(), rows, hr.size() * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hs.data(), scores, hs.size() * 4, hipMemcpyDeviceToHost));
    int found = 0;
    for (int b = 0; b < B; ++b) found += hr[(size_t)b * k] == planted[b] && hs[(size_t)b * k] > 0.99f;
    printf("%.1f us per batch, %.0f GB/s of corpus, %.0f queries/s; planted rows found first: %d / %d\n", ms * 1e3,
           (double)n * ld * 2 / (ms * 1e-3) / 1e9, B / (ms * 1e-3), found, B);
    return found == B ? 0 : 1;
}
