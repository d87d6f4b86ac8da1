// Gather rate of 16-byte loads by alignment (16 / 8 / 4 / 1 bytes) and of 8-byte loads, random addresses inside a
// 1-MB (L2-resident) region -- what a row visit of the solver rounds costs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_align.hip -o /tmp/gather_align && /tmp/gather_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct __attribute__((packed, aligned(1))) U16 { unsigned long long a, b; };
struct __attribute__((packed, aligned(1))) U8 { unsigned long long a; };
template <int MODE>
__global__ void __launch_bounds__(512) k(const unsigned char *p, const unsigned *off, unsigned n, unsigned long long *out) {
    unsigned long long acc = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned o = off[i];
        if (MODE == 0) { const U16 v = *(const U16 *)(p + (o & ~15u)); acc += v.a ^ v.b; }
        if (MODE == 1) { const U16 v = *(const U16 *)(p + (o & ~7u)); acc += v.a ^ v.b; }
        if (MODE == 2) { const U16 v = *(const U16 *)(p + (o & ~3u)); acc += v.a ^ v.b; }
        if (MODE == 3) { const U16 v = *(const U16 *)(p + o); acc += v.a ^ v.b; }
        if (MODE == 4) { const U8 v = *(const U8 *)(p + (o & ~7u)); acc += v.a; }
        if (MODE == 5) { const U16 v = *(const U16 *)(p + (o & ~7u)); const U8 w = *(const U8 *)(p + (o & ~7u) + 16); acc += v.a ^ v.b ^ w.a; }
        if (MODE == 6) { const U16 v = *(const U16 *)(p + (o & ~7u)); const U16 w = *(const U16 *)(p + (o & ~7u) + 16); acc += v.a ^ v.b ^ w.a ^ w.b; }
        if (MODE == 7) { const unsigned v = *(const unsigned *)(p + (o & ~3u)); acc += v; }
    }
    if (acc == 0x1234567ull) out[0] = acc;
}
int main(int argc, char **argv) {
    const unsigned n = 1u << 28, region = argc > 1 ? (unsigned)atoi(argv[1]) : 1u << 20;
    const unsigned share = argc > 2 ? (unsigned)atoi(argv[2]) : 1;     // adjacent lanes that gather the same address
    printf("region %u bytes, %u adjacent lanes share an address\n", region, share);
    std::vector<unsigned> h(n);
    unsigned long long s = 88172645463325252ull;
    for (unsigned i = 0; i < n; ++i) { if (i % share == 0) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; } h[i] = (unsigned)(s % (region - 64)); }
    unsigned char *p; unsigned *off; unsigned long long *out;
    hipMalloc(&p, region); hipMemset(p, 1, region); hipMalloc(&off, 4ull * n); hipMalloc(&out, 8);
    hipMemcpy(off, h.data(), 4ull * n, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"16B aligned 16", "16B aligned 8", "16B aligned 4", "16B aligned 1", "8B aligned 8", "16B+8B aligned 8", "16B+16B aligned 8", "4B aligned 4"};
#define RUN(M) { k<M><<<2048, 512>>>(p, off, n, out); hipEventRecord(e0); for (int r = 0; r < 3; ++r) k<M><<<2048, 512>>>(p, off, n, out); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-20s %.3f ms per %u gathers = %.1f G lane-gathers/s\n", names[M], ms / 3, n, n / (ms / 3) / 1e6); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
