// Does the mapping of an allocation (how large the page-table fragments of its VA -> PA mapping can be) decide how fast a
// random 512-byte-row gather over it runs?  The same gather kernel over a 5.12 GB table obtained (a) by hipMalloc, several times
// (each allocation keeps its predecessors alive, so every one lands somewhere else), (b) through the virtual-memory API with the
// virtual address reserved at 2 MB / 32 MB / 1 GB / 4 GB alignment and ONE physical handle mapped behind it.
//   hipcc --offload-arch=gfx950 -O3 -o exp_tlb exp_tlb.hip && ./exp_tlb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// one 32-lane half-wave reads one 512 B row (float4 per lane); 8 rows in flight per half-wave; rows = hash(counter) % n_rows
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ t, uint32_t n_rows, uint32_t rows_per_group, float* out, uint32_t salt) {
    const uint32_t group = (blockIdx.x * 256 + threadIdx.x) >> 5, lane = threadIdx.x & 31;
    float acc = 0.f;
    for (uint32_t i = 0; i < rows_per_group; i += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t r = mix((group * rows_per_group + i + u) ^ salt) % n_rows;
            v[u] = t[(size_t)r * 32 + lane];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;      // never true: keeps the loads
}

static int measure(const char* what, void* p, size_t bytes, float* out) {
    const uint32_t n_rows = (uint32_t)(bytes / 512), groups = 256 * 8 * 8 * 4, rpg = 512;       // 65536 groups x 512 rows = 17 GB
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 40; ++w) gather<<<groups / 8, 256>>>((const float4*)p, n_rows, rpg, out, w);     // ~1.5 s of load first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int K = 10;
    for (int k = 0; k < K; ++k) gather<<<groups / 8, 256>>>((const float4*)p, n_rows, rpg, out, 1000 + k);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double gb = (double)groups * rpg * 512 / 1e9;
    printf("{\"what\": \"%s\", \"va\": \"%p\", \"va_align_log2\": %d, \"ms\": %.3f, \"GBs\": %.0f}\n", what, p, __builtin_ctzll((unsigned long long)p), ms / K, gb / (ms / K) * 1e3);
    fflush(stdout);
    return 0;
}

int main() {
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)10000001 * 512;
    float* out; CK(hipMalloc(&out, 4096));
    std::vector<void*> keep;
    for (int i = 0; i < 6; ++i) {
        void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
        if (measure("hipMalloc", p, bytes, out)) return 1;
        keep.push_back(p);
        void* pad; CK(hipMalloc(&pad, (size_t)(2 + 6 * i) << 20)); keep.push_back(pad);      // shift the next one
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("{\"granularity\": %zu}\n", gran);
    const size_t size = (bytes + gran - 1) / gran * gran;
    for (size_t align : {(size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30, (size_t)4 << 30}) {
        void* va = nullptr;
        hipError_t e = hipMemAddressReserve(&va, size, align, nullptr, 0);
        if (e != hipSuccess) { printf("{\"reserve_align\": %zu, \"error\": \"%s\"}\n", align, hipGetErrorString(e)); continue; }
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, size, &prop, 0));
        CK(hipMemMap(va, size, 0, h, 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, size, &acc, 1));
        CK(hipMemset(va, 0, bytes));
        char what[64]; snprintf(what, sizeof what, "vmm_align_%zuMB", align >> 20);
        if (measure(what, va, bytes, out)) return 1;
    }
    return 0;
}
