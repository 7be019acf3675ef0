// Measures what the ingest path can hope for on this box: pinned H2D / D2H rates, multi-threaded host memcpy into
// pinned memory, pageable hipMemcpy.   hipcc -O2 -o /tmp/h2d_probe tools/cpp/h2d_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pmemcpy(char *d, const char *s, size_t n, int nt)
{
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=] { size_t a = n * t / nt, b = n * (t + 1) / nt; memcpy(d + a, s + a, b - a); });
    for (auto &x : th) x.join();
}
int main()
{
    const size_t N = (size_t)1200 << 20;
    char *pageable = (char *)malloc(N); memset(pageable, 1, N);
    char *pinned; hipHostMalloc((void **)&pinned, N, hipHostMallocDefault); memset(pinned, 2, N);
    char *dev; hipMalloc((void **)&dev, N);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t1 = now();
        printf("pinned H2D   %6.1f GB/s\n", N / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpyAsync(pinned, dev, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now();
        printf("pinned D2H   %6.1f GB/s\n", N / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpyAsync(dev, pageable, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t1 = now();
        printf("pageable H2D %6.1f GB/s\n", N / (t1 - t0) / 1e9);
        t0 = now(); hipMemcpyAsync(pageable, dev, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now();
        printf("pageable D2H %6.1f GB/s\n", N / (t1 - t0) / 1e9);
    }
    for (int nt : {1, 2, 4, 8, 12, 16}) {
        double best = 0;
        for (int rep = 0; rep < 3; rep++) { double t0 = now(); pmemcpy(pinned, pageable, N, nt); double t1 = now(); double r = N / (t1 - t0) / 1e9; if (r > best) best = r; }
        printf("memcpy pageable->pinned, %2d threads: %6.1f GB/s\n", nt, best);
    }
    // chunked pipeline: 8 threads stage 64 MB chunks into a 3-slot pinned ring while the DMA engine drains it
    for (int nt : {4, 8, 12}) {
        const size_t C = (size_t)64 << 20; const int K = 3;
        hipEvent_t ev[K]; for (int i = 0; i < K; i++) hipEventCreate(&ev[i]);
        double t0 = now();
        for (size_t off = 0, i = 0; off < N; off += C, i++) {
            const size_t len = off + C <= N ? C : N - off; const int sl = i % K;
            if (i >= (size_t)K) hipEventSynchronize(ev[sl]);
            pmemcpy(pinned + sl * C, pageable + off, len, nt);
            hipMemcpyAsync(dev + off, pinned + sl * C, len, hipMemcpyHostToDevice, s); hipEventRecord(ev[sl], s);
        }
        hipStreamSynchronize(s); double t1 = now();
        printf("staged pipeline pageable->device, %2d threads: %6.1f GB/s (%.1f ms)\n", nt, N / (t1 - t0) / 1e9, (t1 - t0) * 1e3);
    }
    printf("hw threads %u\n", std::thread::hardware_concurrency());
    return 0;
}
