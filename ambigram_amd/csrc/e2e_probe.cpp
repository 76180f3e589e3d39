// e2e_probe.cpp -- the single-sample latency as a C caller of the ABI sees it (bench.py runs this next to its own ctypes loop):
//   e2e_probe <sample.lh> <chromosome.sol> [reps]
// One chromosome of one sample; per repetition  ambi_batch_upload -> ambi_batch_run -> ambi_batch_fetch_paths ->
// ambi_batch_unit_path  (the packed unit on the host -> its final path on the host, SURVEY.md 8d's region), then ambi_batch_wait
// outside the timed span.  Prints one JSON object.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/ambigram_hip.h"

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: e2e_probe <lh> <sol> [reps]\n"); return 2; }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const int reps = argc > 3 ? atoi(argv[3]) : 300;
    ambi_graph_t* g = nullptr;
    if (ambi_graph_read_lh(argv[1], &g) != 0) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    ambi_set_device(0);
    ambi_batch_t* b = nullptr;
    ambi_batch_create(&b);
    if (ambi_batch_add_chromosome_sol(b, g, 0, argv[2]) < 0) { fprintf(stderr, "cannot add the unit\n"); return 1; }
    std::vector<int32_t> path(1 << 16);
    int rc = 0, len = 0;
    auto once = [&]() {
        if ((rc = ambi_batch_upload(b)) || (rc = ambi_batch_run(b, 0, nullptr)) || (rc = ambi_batch_fetch_paths(b))) return;
        len = ambi_batch_unit_path(b, 0, 1, path.data(), (int32_t)path.size());
    };
    for (int i = 0; i < 10 && !rc; i++) { once(); ambi_batch_wait(b); }
    double total = 0, best = 1e9;
    for (int i = 0; i < reps && !rc; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        once();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        total += us; if (us < best) best = us;
        ambi_batch_wait(b);
    }
    if (rc) { fprintf(stderr, "engine: %s\n", ambi_error_string(rc)); return 1; }
    long sum = 0;
    for (int i = 0; i < len; i++) sum += path[i] * (long)(i % 7 + 1);
    printf("{\"e2e_us_mean\": %.2f, \"e2e_us_best\": %.2f, \"reps\": %d, \"path_len\": %d, \"path_checksum\": %ld}\n", total / reps, best, reps, len, sum);
    ambi_batch_destroy(b);
    ambi_graph_destroy(g);
    return 0;
}
