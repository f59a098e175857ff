#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
int main() {
    for (int n : {1, 8, 16, 32, 64, 128, 256}) {
        std::atomic<long long> total{0};
        std::vector<std::thread> th;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) th.emplace_back([&] { volatile double x = 1.0; long long c = 0; auto e = std::chrono::steady_clock::now() + std::chrono::milliseconds(300); while (std::chrono::steady_clock::now() < e) { for (int k = 0; k < 1000; k++) x = x * 1.0000001 + 1e-9; c += 1000; } total += c; });
        for (auto& t : th) t.join();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %3d: %.1f M iters/s total, %.2f per thread (wall %.2f s)\n", n, total / dt / 1e6, total / dt / 1e6 / n, dt);
    }
}
