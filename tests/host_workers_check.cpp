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
    int caught = 0;
    try { w.Run([&](int d) { if (d == n - 1) throw MgcgError("boom"); }); } catch (MgcgError& e) { caught = std::string(e.what()) == "boom"; }
    try { w.Run([&](int) {}); caught += 1; } catch (...) { caught = -100; }     // the error does not stick
    printf("ok %d\n", caught);
    return caught == 2 ? 0 : 2;
}
