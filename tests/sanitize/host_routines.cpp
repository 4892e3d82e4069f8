// tests/sanitize/host_routines.cpp -- the HOST routines of libvbx_hip.so under AddressSanitizer + UBSan on the CPU.
//
// GPU AddressSanitizer is not available on the MI355X pool, and the kernels have no CPU build (SURVEY.md section 5, row
// "sanitizers"); what CAN run under the sanitizers is every routine of the library that never touches the device -- the
// average linkage in both arithmetic forms and the flat-cluster cut (vbhmm.py:140-146), and the Kaldi archive index
// (vbhmm.py:117) -- compiled from the very header the library includes (vbx_amd/csrc/vbx_linkage.hpp), with g++:
//
//     g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I vbx_amd/csrc \
//         -o /tmp/host_routines tests/sanitize/host_routines.cpp && /tmp/host_routines
//
// tests/test_sanitize.py builds and runs it (-m "not gpu").  Inputs: random and adversarial (massive ties, n = 1, 2, duplicate
// points; truncated, empty and garbage archives; output arrays that are exactly long enough or too short).  Beyond "no
// sanitizer report" it checks the invariants a linkage matrix must have.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "vbx_linkage.hpp"

static void check_linkage(int64_t n, const std::vector<double>& Z) {
    // row k merges two existing clusters into cluster n + k; heights are non-decreasing; sizes add up; the last row holds all points
    std::vector<int64_t> size((size_t)(2 * n), 1);
    double prev = -1e300;
    for (int64_t k = 0; k + 1 < n; ++k) {
        const double a = Z[4 * k], b = Z[4 * k + 1], d = Z[4 * k + 2], m = Z[4 * k + 3];
        assert(a >= 0 && b > a && b < (double)(n + k));
        assert(d >= prev);
        prev = d;
        size[(size_t)(n + k)] = size[(size_t)a] + size[(size_t)b];
        assert((double)size[(size_t)(n + k)] == m);
    }
    if (n > 1) assert(Z[4 * (n - 2) + 3] == (double)n);
}

static void linkage_cases(std::mt19937_64& rng) {
    std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int64_t n : {1, 2, 3, 7, 64, 257, 600}) {
        for (int kind = 0; kind < 4; ++kind) {
            // kind 0: random distances; 1: points on a line (many near-ties); 2: all equal (massive exact ties); 3: two tight clumps
            std::vector<double> cond((size_t)(n * (n - 1) / 2 + 1));
            std::vector<double> x((size_t)n);
            for (auto& v : x) v = kind == 3 ? (U(rng) < 0.5 ? 0.0 : 5.0) + 1e-3 * U(rng) : std::floor(10 * U(rng));
            size_t p = 0;
            for (int64_t i = 0; i < n; ++i)
                for (int64_t j = i + 1; j < n; ++j)
                    cond[p++] = kind == 0 ? U(rng) : kind == 2 ? 1.0 : std::fabs(x[(size_t)i] - x[(size_t)j]);
            cond.resize(p);                                      // EXACTLY n (n - 1) / 2 entries: a read past the end is a report
            std::vector<double> Z((size_t)(4 * (n > 1 ? n - 1 : 1))), Zf(Z.size());
            Z.resize((size_t)(4 * (n - 1)));
            Zf.resize(Z.size());
            vbx::average_linkage(n, cond.data(), Z.data());
            vbx::average_linkage_fastcluster(n, cond.data(), Zf.data());
            check_linkage(n, Z);
            check_linkage(n, Zf);
            for (double t : {-1.0, 0.0, 0.5, 1.0, 1e9}) {
                std::vector<int32_t> lab((size_t)n);
                vbx::fcluster_distance(n, Z.data(), t, lab.data());
                int32_t mx = 0;
                for (int32_t l : lab) { assert(l >= 1 && l <= n); mx = std::max(mx, l); }
                if (t >= 1e9) assert(mx == 1);                   // everything merges below a huge threshold
                if (t < 0 && n > 1 && kind == 0) assert(mx == n); // nothing merges below a negative one
            }
        }
    }
}

static std::string entry(const std::string& key, int32_t dim, int es, std::mt19937_64& rng) {
    std::string s = key + " ";
    s.push_back('\0'); s.push_back('B');
    s += es == 4 ? "FV " : "DV ";
    s.push_back('\4');
    s.append(reinterpret_cast<const char*>(&dim), 4);
    for (int i = 0; i < dim * es; ++i) s.push_back((char)(rng() & 0xff));
    return s;
}

static void archive_cases(std::mt19937_64& rng) {
    std::string ark;
    const int n = 37;
    for (int k = 0; k < n; ++k) ark += entry("rec" + std::to_string(k % 5) + "_" + std::to_string(k), k % 3 == 0 ? 0 : 16 + k, k % 2 ? 4 : 8, rng);
    auto run = [&](const std::string& buf, int64_t cap) {
        // the buffer is copied into an allocation of exactly its size, the outputs are exactly `cap` long
        std::vector<unsigned char> b(buf.begin(), buf.end());
        std::vector<int64_t> ko((size_t)cap), dof((size_t)cap);
        std::vector<int32_t> kl((size_t)cap), dm((size_t)cap), es((size_t)cap);
        return vbx::ark_index(b.data(), (int64_t)b.size(), cap, ko.data(), kl.data(), dof.data(), dm.data(), es.data());
    };
    assert(run(ark, n) == n);
    assert(run(ark, n + 10) == n);
    assert(run(ark, n - 1) == -2);                               // too little room: asked to come back
    assert(run(ark, 0) == -2);
    assert(run(std::string(), 4) == 0);
    for (size_t cut = 1; cut < ark.size(); cut += 7) {           // every truncation is either a shorter archive or "malformed", never a read past the end
        const int64_t r = run(ark.substr(0, cut), n);
        assert(r == -1 || (r >= 0 && r <= n));
    }
    for (int rep = 0; rep < 200; ++rep) {                        // garbage and bit flips
        std::string g = ark;
        for (int f = 0; f < 1 + rep % 5; ++f) g[(size_t)(rng() % g.size())] = (char)(rng() & 0xff);
        const int64_t r = run(g, n + 5);
        assert(r >= -2 && r <= n + 5);
    }
    std::string text = "rec0_1  [ 1.0 2.0 3.0 ]\n";
    assert(run(text, 4) == -1);                                  // a text archive is not this routine's business
}

int main() {
    std::mt19937_64 rng(12345);
    linkage_cases(rng);
    archive_cases(rng);
    std::puts("host routines: OK under the sanitizers");
    return 0;
}
