// cost of HIP event objects on the host: create / record / synchronize / elapsed / destroy (tools/exp: measured to decide on an event pool in dev.h)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const int N = 2000;
    hipEvent_t e[2];
    double tc = 0, tr = 0, tw = 0, te = 0, td = 0;
    for (int i = 0; i < N; i++) {
        auto t0 = now(); hipEventCreate(&e[0]); hipEventCreate(&e[1]);
        auto t1 = now(); hipEventRecord(e[0], s); hipEventRecord(e[1], s);
        auto t2 = now(); hipEventSynchronize(e[1]);
        auto t3 = now(); float ms; hipEventElapsedTime(&ms, e[0], e[1]);
        auto t4 = now(); hipEventDestroy(e[0]); hipEventDestroy(e[1]);
        auto t5 = now();
        tc += us(t0, t1); tr += us(t1, t2); tw += us(t2, t3); te += us(t3, t4); td += us(t4, t5);
    }
    printf("per pair of events (us): create %.2f, record %.2f, synchronize %.2f, elapsed %.2f, destroy %.2f\n", tc / N, tr / N, tw / N, te / N, td / N);
    // re-used events
    hipEventCreate(&e[0]); hipEventCreate(&e[1]); tr = tw = te = 0;
    for (int i = 0; i < N; i++) {
        auto t1 = now(); hipEventRecord(e[0], s); hipEventRecord(e[1], s);
        auto t2 = now(); hipEventSynchronize(e[1]);
        auto t3 = now(); float ms; hipEventElapsedTime(&ms, e[0], e[1]);
        auto t4 = now();
        tr += us(t1, t2); tw += us(t2, t3); te += us(t3, t4);
    }
    printf("re-used pair (us): record %.2f, synchronize %.2f, elapsed %.2f\n", tr / N, tw / N, te / N);
    return 0;
}
