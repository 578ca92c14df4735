// hipMalloc / hipFree / first-touch cost by size on the GPU box (the fused search's first device call pays for the working
// buffers): build with hipcc --offload-arch=gfx950 -O2 -o malloc_probe malloc_probe.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void touch(unsigned *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1u;
}

int main() {
    double t0 = now();
    hipSetDevice(0);
    hipFree(nullptr);
    printf("context %.3f s\n", now() - t0);
    t0 = now();
    hipStream_t s;
    hipStreamCreate(&s);
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, (unsigned *)nullptr, (size_t)0);
    hipStreamSynchronize(s);
    printf("first launch (code object load) %.3f s\n", now() - t0);
    const size_t sizes[] = {64ull << 20, 1ull << 30, 4ull << 30, 16ull << 30, 32ull << 30};
    for (size_t sz : sizes) {
        void *p = nullptr;
        t0 = now();
        hipError_t e = hipMalloc(&p, sz);
        const double tm = now() - t0;
        if (e != hipSuccess) { printf("%zu MB: malloc failed\n", sz >> 20); continue; }
        t0 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (unsigned *)p, sz / 4);
        hipStreamSynchronize(s);
        const double t1 = now() - t0;
        t0 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (unsigned *)p, sz / 4);
        hipStreamSynchronize(s);
        const double t2 = now() - t0;
        t0 = now();
        hipFree(p);
        const double tf = now() - t0;
        printf("%6zu MB: hipMalloc %.4f s, first touch %.4f s, second touch %.4f s, hipFree %.4f s\n", sz >> 20, tm, t1, t2, tf);
    }
    // many medium allocations (a batch's ~40 buffers)
    t0 = now();
    std::vector<void *> v;
    for (int i = 0; i < 40; i++) { void *p; hipMalloc(&p, 256ull << 20); v.push_back(p); }
    printf("40 x 256 MB hipMalloc %.4f s\n", now() - t0);
    t0 = now();
    for (void *p : v) hipFree(p);
    printf("40 x hipFree %.4f s\n", now() - t0);
    // pinned host memory and copies
    const size_t hb = 280ull << 20;
    void *h = malloc(hb), *hp = nullptr, *d = nullptr;
    memset(h, 1, hb);
    hipMalloc(&d, hb);
    t0 = now(); hipMemcpy(d, h, hb, hipMemcpyHostToDevice); printf("280 MB pageable H2D %.4f s\n", now() - t0);
    t0 = now(); hipMemcpy(d, h, hb, hipMemcpyHostToDevice); printf("280 MB pageable H2D again %.4f s\n", now() - t0);
    t0 = now(); hipHostMalloc(&hp, hb, 0); printf("hipHostMalloc 280 MB %.4f s\n", now() - t0);
    memcpy(hp, h, hb);
    t0 = now(); hipMemcpy(d, hp, hb, hipMemcpyHostToDevice); printf("280 MB pinned H2D %.4f s\n", now() - t0);
    t0 = now(); hipHostRegister(h, hb, 0); printf("hipHostRegister 280 MB %.4f s\n", now() - t0);
    t0 = now(); hipMemcpy(d, h, hb, hipMemcpyHostToDevice); printf("280 MB registered H2D %.4f s\n", now() - t0);
    return 0;
}
