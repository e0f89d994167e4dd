// lds_gather — what does a FAST ring gather cost in LDS?  (round 6; build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_gather.hip -o tools/micro/lds_gather)
// Every lane gathers the 16 ring pixels + centre of a pseudo-random pixel of a 40 x 272-byte tile (k_fast_rows' corner pass), as
//   A: 17 ds_read_u8                                   (round 5)
//   B: 7 ds_read_b64 at BYTE-unaligned addresses x-3   (one per ring row; ROCm runs gfx9 with SH_MEM_CONFIG.ALIGNMENT_MODE = UNALIGNED)
//   C: 7 rows as ds_read_b32 pairs at x-3, x+1 (unaligned dwords)
// 12 single-wave workgroups per CU, 3072 workgroups; prints ns per gather-iteration per wave and checks that A, B, C return the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define PITCH 272
struct __attribute__((packed, aligned(1))) U64 { unsigned long long v; };
struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, uint32_t* out)
{
    extern __shared__ uint8_t t[];
    for (int i = threadIdx.x; i < 40 * PITCH; i += 64) t[i] = (uint8_t)(i * 31 + (i >> 8) * 7 + blockIdx.x);
    __syncthreads();
    uint32_t s = blockIdx.x * 64 + threadIdx.x + 1, acc = 0;
    constexpr int RO[16] = { 3 * PITCH + 0, 3 * PITCH + 1, 2 * PITCH + 2, 1 * PITCH + 3, 0 * PITCH + 3, -1 * PITCH + 3, -2 * PITCH + 2, -3 * PITCH + 1,
                             -3 * PITCH + 0, -3 * PITCH - 1, -2 * PITCH - 2, -1 * PITCH - 3, 0 * PITCH - 3, 1 * PITCH - 3, 2 * PITCH - 2, 3 * PITCH - 1 };
    for (int it = 0; it < iters; it++) {
        // neighbouring lanes take neighbouring pixels most of the time (edges), as the corner pass does
        const uint32_t r = rnd(s);
        const int y = 3 + (int)(r % 34), x = 3 + (int)((r >> 8) % 250);
        const uint8_t* c = t + y * PITCH + x;
        uint32_t ring[16], v;
        if (MODE == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) ring[k2] = c[RO[k2]];
            v = c[0];
        } else if (MODE == 1) {
            unsigned long long row[7];
#pragma unroll
            for (int d = 0; d < 7; d++) row[d] = reinterpret_cast<const U64*>(c + (d - 3) * PITCH - 3)->v;
            auto px = [&](int dy, int dx) { return (uint32_t)(row[dy + 3] >> (8 * (dx + 3))) & 0xFFu; };
            const int DX[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 }, DY[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) ring[k2] = px(DY[k2], DX[k2]);
            v = px(0, 0);
        } else if (MODE >= 3) {
            // inline asm: exactly these instructions.  3: 7 x ds_read_b64 unaligned; 4: 14 x ds_read_b32 unaligned; 5: 7 x ds_read_b64 at addresses rounded down to 8 (aligned: not the same bytes)
            unsigned long long row[7];
            const uint32_t a0 = (uint32_t)(uintptr_t)(c - 3 * PITCH - 3) - (uint32_t)(uintptr_t)t;
#pragma unroll
            for (int d = 0; d < 7; d++) {
                const uint32_t a = a0 + d * PITCH;
                if (MODE == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(row[d]) : "v"(a));
                else if (MODE == 5) asm volatile("ds_read_b64 %0, %1" : "=v"(row[d]) : "v"(a & ~7u));
                else { uint32_t lo, hi; asm volatile("ds_read_b32 %0, %1" : "=v"(lo) : "v"(a)); asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(hi) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); row[d] = lo | ((unsigned long long)hi << 32); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            auto px = [&](int dy, int dx) { return (uint32_t)(row[dy + 3] >> (8 * (dx + 3))) & 0xFFu; };
            const int DX[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 }, DY[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) ring[k2] = px(DY[k2], DX[k2]);
            v = px(0, 0);
        } else {
            uint32_t lo[7], hi[7];
#pragma unroll
            for (int d = 0; d < 7; d++) { lo[d] = reinterpret_cast<const U32*>(c + (d - 3) * PITCH - 3)->v; hi[d] = reinterpret_cast<const U32*>(c + (d - 3) * PITCH + 1)->v; }
            auto px = [&](int dy, int dx) { const int b = dx + 3; return ((b < 4 ? lo[dy + 3] : hi[dy + 3]) >> (8 * (b & 3))) & 0xFFu; };
            const int DX[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 }, DY[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) ring[k2] = px(DY[k2], DX[k2]);
            v = px(0, 0);
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) acc = acc * 3u + ring[k2];
        acc += v * 17u;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main()
{
    const int nb = 3072, iters = 2000;
    uint32_t* d[6];
    std::vector<uint32_t> h[6];
    for (int m = 0; m < 6; m++) { hipMalloc(&d[m], nb * 64 * 4); h[m].resize(nb * 64); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* name[6] = { "17 x ds_read_u8", "7 x 8 bytes, compiler (3 read2_b64 + 1 b64, unaligned)", "14 x 4 bytes, compiler (same)", "7 x ds_read_b64 unaligned (asm)", "14 x ds_read_b32 unaligned (asm)", "7 x ds_read_b64 ALIGNED (asm; other bytes)" };
    for (int m = 0; m < 6; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(nb), dim3(64), 12800, 0, iters, d[m]);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-56s %8.3f ms  = %6.1f ns per gather iteration and wave (12 waves per CU)\n", name[m], ms, ms * 1e6 / iters);
        }
        hipMemcpy(h[m].data(), d[m], nb * 64 * 4, hipMemcpyDeviceToHost);
    }
    int bad = 0;
    int bad_asm[2] = { 0, 0 };
    for (int i = 0; i < nb * 64; i++) { bad += (h[0][i] != h[1][i]) + (h[0][i] != h[2][i]); bad_asm[0] += h[0][i] != h[3][i]; bad_asm[1] += h[0][i] != h[4][i]; }
    printf("%s: the compiler's unaligned LDS reads return the same bytes as byte reads in %d lanes x %d iterations\n", bad ? "MISMATCH" : "OK", nb * 64, iters);
    printf("(asm variants, timing only: %d / %d lanes differ)\n", bad_asm[0], bad_asm[1]);
    return bad != 0;
}
