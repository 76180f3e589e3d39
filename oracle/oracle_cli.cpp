// oracle/oracle_cli.cpp -- TEST INFRASTRUCTURE ONLY.
// Command-line front end of the CPU oracle: same flags as the reference CLI (localhap.cpp:22-40) plus
// --sol a.sol[,b.sol...] which stands in for the external `cbc` call (localhap.cpp:179-181).
// Prints the stdout lines the reference would print (solver chatter excluded).
#include <cstring>
#include <iostream>
#include <string>
#include "bfb_oracle.hpp"

int main(int argc, char** argv) {
    oracle::RunOptions opt;
    std::string sols;
    auto truthy = [](const char* s) { return !strcmp(s, "true") || !strcmp(s, "1"); };
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i]; const char* v = argv[i + 1];
        if (k == "--in_lh") opt.lh = v;
        else if (k == "--sol") sols = v;
        else if (k == "--juncdb") opt.juncs = v;
        else if (k == "--junc_info") opt.juncInfo = truthy(v);
        else if (k == "--reversed") opt.reversed = truthy(v);
        else if (k == "--all") opt.all = truthy(v);
        else if (k == "--lp_prefix") opt.lpPrefix = v;
        else if (k == "--op") { if (strcmp(v, "bfb")) { std::cerr << "only --op bfb\n"; return 2; } }
    }
    size_t p = 0;
    while (!sols.empty()) { size_t q = sols.find(',', p); opt.solPerChr.push_back(sols.substr(p, q - p)); if (q == std::string::npos) break; p = q + 1; }
    oracle::RunResult R = oracle::runBfb(opt);
    for (auto& l : R.log) std::cout << l << "\n";
    if (!R.ok) { std::cerr << R.err << "\n"; return 1; }
    return 0;
}
