// Micro-benchmark (GPU box): what bounds the bucket scatter of the row build?
// Records arrive genome by genome (record i = genome i / nb, bucket i % nb) and go
// to slot bstart[bucket] + rank -- the S4 pattern.  hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill(uint4 *rec, u32 *rank, u32 n, u32 nb) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rec[i] = make_uint4(i, i + 200, i / nb, i % nb);
    rank[i] = i / nb;
}
__global__ void fill_bstart(u32 *bstart, u32 nb, u32 per) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nb) bstart[i] = i * per;
}
// mode 0: as in the product; 1: no bstart gather; 2: XCD-contiguous blocks; 3: 8-bucket transposed lines;
// 4: read only (no write); 5: 4-byte write; 6: sequential write (copy); 7: as 0 with non-temporal stores
__global__ void __launch_bounds__(256) scatter(const uint4 *__restrict__ rec, const u32 *__restrict__ rank,
                                               const u32 *__restrict__ bstart, uint4 *__restrict__ S, u32 n, u32 per, int mode) {
    u32 blk = blockIdx.x;
    if (mode == 2) blk = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const u32 d = blk * 256 + threadIdx.x;
    if (d >= n) return;
    const u32 rk = rank[d];
    const uint4 r = rec[d];
    u32 slot;
    if (mode == 1 || mode == 5) slot = r.w * per + rk;
    else if (mode == 3) slot = (r.w >> 3) * (per * 8) + rk * 8 + (r.w & 7);
    else if (mode == 6) slot = d;
    else slot = bstart[r.w] + rk;
    if (mode == 4) { if (slot == 0xffffffffu) S[0] = r; return; }
    if (mode == 5) { ((u32 *)S)[slot] = r.x; return; }
    if (mode == 7) {
        u32 *q = (u32 *)(S + slot);
        __builtin_nontemporal_store(r.x, q); __builtin_nontemporal_store(r.y, q + 1);
        __builtin_nontemporal_store(r.z, q + 2); __builtin_nontemporal_store(r.w, q + 3);
        return;
    }
    S[slot] = r;
}
int main() {
    const u32 nb = 4u << 20, per = 64, n = nb * per;   // 268 M records
    uint4 *rec, *S; u32 *rank, *bstart;
    CK(hipMalloc(&rec, 16ull * n)); CK(hipMalloc(&S, 16ull * n)); CK(hipMalloc(&rank, 4ull * n)); CK(hipMalloc(&bstart, 4ull * (nb + 1)));
    fill<<<n / 256, 256>>>(rec, rank, n, nb);
    fill_bstart<<<nb / 256 + 1, 256>>>(bstart, nb, per);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"product", "no bstart gather", "XCD-contiguous", "8-bucket transposed lines", "read only", "4-byte write", "sequential copy", "non-temporal stores"};
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            scatter<<<(n / 256 + 7) / 8 * 8, 256>>>(rec, rank, bstart, S, n, per, mode);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("mode %d %-28s %7.3f ms  %6.1f G rec/s  %6.2f TB/s (36 B/rec)\n", mode, names[mode], best, n / best / 1e6, 36.0 * n / best / 1e9);
    }
    return 0;
}
