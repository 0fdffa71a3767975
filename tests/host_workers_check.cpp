// DeviceWorkers (host/Mgcg.hpp) under ThreadSanitizer: phases on n long-lived workers, results read by the caller between phases,
// an exception in one worker surfaces in Run() and does not stick.  Built and run by tests/test_host_logic.py (no GPU).
#include "Mgcg.hpp"
#include <cstdio>
using namespace LWisteria::Mgcg;
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8, phases = argc > 2 ? atoi(argv[2]) : 20000;
    DeviceWorkers w(n);
    std::vector<long long> acc((size_t)n, 0);
    long long shared = 0;
    for (int p = 0; p < phases; p++) {
        w.Run([&](int d) { acc[(size_t)d] += d + p; });
        for (int d = 0; d < n; d++) shared += acc[(size_t)d];          // the caller reads what the workers wrote: Run() must order it
        if (p % 1000 == 999) std::this_thread::sleep_for(std::chrono::microseconds(300));   // lets the workers fall asleep on the condition variable
    }
    long long expect = 0, run = 0;
    std::vector<long long> a2((size_t)n, 0);
    for (int p = 0; p < phases; p++) { for (int d = 0; d < n; d++) { a2[(size_t)d] += d + p; } for (int d = 0; d < n; d++) expect += a2[(size_t)d]; }
    (void)run;
    if (shared != expect) { printf("MISMATCH %lld %lld\n", shared, expect); return 1; }
    // The native multi-device Solve() of host/Mgcg.hpp: ONE phase whose n bodies are in flight together and meet in collectives
    // (here: a barrier with a rank-order sum, the shape of the loopback all-reduce), each body leaves its (iteration, residual) in its
    // own slot and the caller reads slot 0 afterwards.  All n bodies must run concurrently or this never returns.
    {
        std::mutex bm; std::condition_variable bcv; int arrived = 0; long long gen = 0;
        std::vector<double> slot((size_t)n, 0.0), result((size_t)n, 0.0);
        std::vector<int> iteration((size_t)n, -1);
        auto barrier = [&] { std::unique_lock<std::mutex> lk(bm); const long long g = gen; if (++arrived == n) { arrived = 0; ++gen; bcv.notify_all(); } else bcv.wait(lk, [&] { return gen != g; }); };
        for (int rep = 0; rep < 50; rep++) {
            w.Run([&](int d) {
                double rr = 0;
                for (int it = 0; it < 20; it++) {
                    slot[(size_t)d] = d + it + rep;
                    barrier();
                    double sum = 0; for (int q = 0; q < n; q++) sum += slot[(size_t)q];
                    barrier();
                    rr = sum; iteration[(size_t)d] = it;
                }
                result[(size_t)d] = rr;
            });
            double want = 0; for (int q = 0; q < n; q++) want += q + 19 + rep;
            for (int d = 0; d < n; d++) if (result[(size_t)d] != want || iteration[(size_t)d] != 19) { printf("COLLECTIVE MISMATCH\n"); return 3; }
        }
    }
    int caught = 0;
    try { w.Run([&](int d) { if (d == n - 1) throw MgcgError("boom"); }); } catch (MgcgError& e) { caught = std::string(e.what()) == "boom"; }
    try { w.Run([&](int) {}); caught += 1; } catch (...) { caught = -100; }     // the error does not stick
    printf("ok %d\n", caught);
    return caught == 2 ? 0 : 2;
}
