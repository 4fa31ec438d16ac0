// Candidate placement probes (tools/exp_probe.py): which synthetic access pattern separates the allocations on which the
// headline launch runs 412 us from those on which it runs 470 us?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libprobe.so probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void stream_write(float4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// the output pattern of the fused forward: one wave per tile of 64 elements; per tile 512 B of ids, 3 x 256 B of floats, and
// 4 x 4 B into per-query arrays; with `table` also 64 random 512 B row reads per tile (two per lane pair...: 32 lanes x float4 = a row)
__global__ __launch_bounds__(256) void tile_pattern(char* base, uint32_t n_tiles, const float4* table, uint32_t n_rows, uint32_t salt) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = gridDim.x * 4;
    char* ids = base;
    char* f0 = ids + (size_t)n_tiles * 512;
    char* f1 = f0 + (size_t)n_tiles * 256;
    char* f2 = f1 + (size_t)n_tiles * 256;
    char* s0 = f2 + (size_t)n_tiles * 256;
    for (uint32_t t = wave; t < n_tiles; t += n_waves) {
        float acc = 0.f;
        if (table) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {            // 2 half-waves x 16 x 2 passes = 64 rows per tile
                const uint32_t r = mix((t * 64 + u * 2 + (lane >> 5)) ^ salt) % n_rows;
                v[u] = table[(size_t)r * 32 + (lane & 31)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u].x + v[u].w;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const uint32_t r = mix((t * 64 + 32 + u * 2 + (lane >> 5)) ^ salt) % n_rows;
                v[u] = table[(size_t)r * 32 + (lane & 31)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u].y + v[u].z;
        }
        ((int64_t*)(ids + (size_t)t * 512))[lane] = (int64_t)lane + (int64_t)acc;
        ((float*)(f0 + (size_t)t * 256))[lane] = acc;
        ((float*)(f1 + (size_t)t * 256))[lane] = acc + 1.f;
        ((float*)(f2 + (size_t)t * 256))[lane] = acc + 2.f;
        if (lane < 4) ((float*)(s0 + (size_t)lane * n_tiles * 4))[t] = acc;
    }
}

extern "C" int probe_stream(void* p, size_t bytes, void* stream) {
    stream_write<<<2048, 256, 0, (hipStream_t)stream>>>((float4*)p, bytes / 16);
    return (int)hipGetLastError();
}
extern "C" int probe_tiles(void* p, uint32_t n_tiles, const void* table, uint32_t n_rows, uint32_t salt, int blocks, void* stream) {
    tile_pattern<<<blocks, 256, 0, (hipStream_t)stream>>>((char*)p, n_tiles, (const float4*)table, n_rows, salt);
    return (int)hipGetLastError();
}
