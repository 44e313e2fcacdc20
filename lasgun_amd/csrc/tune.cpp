// lasgun_amd/csrc/tune.cpp -- see tune.h.  (Round 6: moved out of capi.cpp; the race no longer fails the caller's render when a
// candidate cannot run, counts API calls instead of launches, and its table can be exported, imported and cleared.)
#include "tune.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace lg {
namespace tune {

bool Key::operator<(const Key &o) const { return std::lexicographical_compare(v, v + 12, o.v, o.v + 12); }

namespace {
std::mutex g_mtx;     // the two tables (and the file, below)
std::mutex g_run_mtx; // one race at a time in the process: two accels of one kind measuring side by side would time each other
// never destroyed: launches at interpreter exit, after static destructors have begun, still find them
std::map<Key, int> &g_tuned = *new std::map<Key, int>();
std::map<Key, uint64_t> &g_first_seen = *new std::map<Key, uint64_t>(); // the API call a kind was first launched in
// LASGUN_TUNE_FILE=<path>: the table persists across processes -- read once before the first look-up, rewritten (temp file + rename) after every
// choice that is remembered.  A program that renders one frame and exits never measures (mode 1); with a file left by an earlier run it takes the
// measured choice at no cost.  One line per kind: twelve hexadecimal words and the choice.  A file that cannot be read or parsed is ignored.
bool g_file_loaded = false;
const char *tune_file() {
    static const char *path = [] { const char *e = std::getenv("LASGUN_TUNE_FILE"); return e && e[0] ? e : nullptr; }();
    return path;
}
void load_file_locked() { // caller holds g_mtx
    if (g_file_loaded) return;
    g_file_loaded = true;
    const char *path = tune_file();
    if (!path) return;
    std::FILE *f = std::fopen(path, "r");
    if (!f) return;
    char line[512];
    while (std::fgets(line, sizeof line, f)) {
        Key k{};
        int choice = -1, used = 0, n = 0;
        const char *p = line;
        bool ok = true;
        for (int i = 0; i < 12 && ok; ++i) {
            unsigned long long v = 0;
            ok = std::sscanf(p, "%llx%n", &v, &n) == 1;
            k.v[i] = v; p += ok ? n : 0;
        }
        ok = ok && std::sscanf(p, "%d%n", &choice, &used) == 1 && choice >= 0 && choice < 256 && (choice & 15) <= 2;
        if (ok) g_tuned.emplace(k, choice); // (a choice made in this process already wins)
    }
    std::fclose(f);
}
void save_file_locked() { // caller holds g_mtx
    const char *path = tune_file();
    if (!path) return;
    const std::string tmp = std::string(path) + ".tmp";
    std::FILE *f = std::fopen(tmp.c_str(), "w");
    if (!f) return;
    for (const auto &kv : g_tuned) {
        for (int i = 0; i < 12; ++i) std::fprintf(f, "%llx ", (unsigned long long)kv.first.v[i]);
        std::fprintf(f, "%d\n", kv.second);
    }
    const bool ok = std::fclose(f) == 0;
    if (ok) (void)std::rename(tmp.c_str(), path);
    else (void)std::remove(tmp.c_str());
}
} // namespace

int mode() {
    static const int m = [] { const char *e = std::getenv("LASGUN_AUTOTUNE"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();
    return m;
}

bool lookup(const Key &key, int *choice) {
    std::lock_guard<std::mutex> g(g_mtx);
    load_file_locked();
    auto it = g_tuned.find(key);
    if (it == g_tuned.end()) return false;
    *choice = it->second;
    return true;
}

bool first_call_of_kind(const Key &key, uint64_t serial) {
    std::lock_guard<std::mutex> g(g_mtx);
    auto it = g_first_seen.find(key);
    if (it == g_first_seen.end()) { g_first_seen.emplace(key, serial); return true; }
    return it->second == serial;
}

void remember(const Key &key, int choice) {
    std::lock_guard<std::mutex> g(g_mtx);
    load_file_locked();
    g_tuned[key] = choice;
    save_file_locked();
}

size_t snapshot(Key *keys, int *choices, size_t capacity) {
    std::lock_guard<std::mutex> g(g_mtx);
    load_file_locked();
    size_t i = 0;
    for (const auto &kv : g_tuned) {
        if (i < capacity) { if (keys) keys[i] = kv.first; if (choices) choices[i] = kv.second; }
        ++i;
    }
    return i;
}

void clear() {
    std::lock_guard<std::mutex> g(g_mtx);
    g_file_loaded = true; // (what the file held is forgotten too)
    g_tuned.clear();
    g_first_seen.clear();
    save_file_locked();
}

int race(const Key &key, Candidate *cands, int n, int rule, hipStream_t stream, const std::function<void(int)> &launch, float *best_out) {
    std::lock_guard<std::mutex> run(g_run_mtx);
    {   // (another accel of this kind may have measured while this one waited)
        int known;
        if (lookup(key, &known)) return known;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { // no events, no race
        (void)hipGetLastError();
        if (e0) (void)hipEventDestroy(e0);
        return cands[rule].choice;
    }
    std::vector<float> best_ms((size_t)n, INFINITY);
    // pass 0 warms every candidate up (buffers, code, clocks); passes 1-3 time them IN TURN, so that a drift of the clocks or a
    // neighbour's launch hits all alike, and the best of the three counts (the persistent kernels' own run-to-run spread is ~5 %:
    // config 4m's megakernel / queue pair, 6 % apart, was called wrongly by one warm-up + best of two in a row); a candidate that
    // is 1.3 x behind after a pass is out, and launches of a quarter second measure themselves in one pass.  A launch of 20 ms and more is
    // timed twice, and 1.1 x behind is out after the first time: its spread is a per cent or two, and seven candidates of 50 ms three
    // times over were a second and a half of the caller's time (config 5: 1.68 s -> ~0.8 s for the same choice)
    for (int pass = 0; pass < 4; ++pass) {
        float fastest = INFINITY;
        for (int k = 0; k < n; ++k) {
            if (!cands[k].in_race) continue;
            float ms = 0.0f;
            bool ok = hipEventRecord(e0, stream) == hipSuccess;
            if (ok) {
                try { launch(k); } catch (...) { ok = false; } // (out of memory for this organisation's buffers, a refused launch: the others may still fit)
            }
            ok = ok && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError(); // the failed call's sticky "last error" is not the caller's
                cands[k].in_race = false;
                best_ms[k] = INFINITY;
                continue;
            }
            if ((pass > 0 || ms > 250.0f) && ms < best_ms[k]) best_ms[k] = ms;
            fastest = std::min(fastest, pass > 0 || ms > 250.0f ? best_ms[k] : ms);
        }
        int left = 0;
        const bool long_launch = fastest > 20.0f;
        for (int k = 0; k < n; ++k) {
            if (cands[k].in_race && pass > 0 && best_ms[k] > (long_launch ? 1.1f : 1.3f) * fastest) cands[k].in_race = false;
            left += cands[k].in_race ? 1 : 0;
        }
        if (left == 0) break;
        if (left <= 1 && pass > 0) break;
        if (fastest > 250.0f || (long_launch && pass >= 2)) break;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    int best = -1;
    if (std::isfinite(best_ms[rule])) best = rule;
    for (int k = 0; k < n; ++k) {
        if (!std::isfinite(best_ms[k])) continue;
        if (best < 0 || best_ms[k] < best_ms[best] * 0.99f) best = k; // (the rule's choice unless another beats it by 1 %: equal candidates do not flip from run to run)
    }
    if (best_out) for (int k = 0; k < n; ++k) best_out[k] = best_ms[k];
    if (best < 0) return cands[rule].choice; // nothing could be timed: the rule's choice, and the next launch of the kind tries again
    remember(key, cands[best].choice);
    return cands[best].choice;
}

} // namespace tune
} // namespace lg
