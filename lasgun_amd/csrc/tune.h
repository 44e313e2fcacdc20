// lasgun_amd/csrc/tune.h -- the measured choice of a launch's organisation, as a unit of its own: the table of
// remembered choices, the race that fills it, and its export / import (lg_tune_export / lg_tune_import, include/lasgun_hip.h).
//
// The tuner knows nothing about scenes or kernels: a KIND of launch is twelve words the caller makes (launch.cpp: tune_key), a CANDIDATE
// an integer the caller can decode, and a candidate is run by a callback.  Every organisation renders the same bytes (the parity tests
// hold them to that), so what is chosen here never changes a pixel -- only when it arrives.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <functional>

namespace lg {
namespace tune {

struct Key {
    uint64_t v[12];
    bool operator<(const Key &o) const;
};
struct Candidate {
    int choice;   // what is remembered when this candidate wins
    bool in_race; // can take the launch at all
};

// LASGUN_AUTOTUNE: 0 = never measure (the fitted rule), 1 (default) = measure a kind at the second API CALL that launches it, 2 = at the first.
int mode();
// a remembered (measured or imported) choice of the kind
bool lookup(const Key &key, int *choice);
// true while `serial` is the API call in which the kind was first seen (mode 1: that call gets the rule's choice at no cost --
// however many launches of the kind it makes: a banded capture's four bands, two shares of one device -- and the next call measures)
bool first_call_of_kind(const Key &key, uint64_t serial);
// Race the candidates on `stream` and remember the winner: a warm-up pass, then up to three timed passes over the candidates IN TURN
// (HIP events, the host waiting; best of each), `rule` (an index into cands) unless another beats it by 1 %.  A candidate whose launch throws
// is out of the race (the HIP error state is cleared); if none is left the rule's choice is returned and nothing is remembered.  One
// race at a time in the process.  best_ms (n floats, may be null) receives the best time of each candidate, +inf where it has none.
int race(const Key &key, Candidate *cands, int n, int rule, hipStream_t stream, const std::function<void(int)> &launch, float *best_ms);
// pin a choice (lg_tune_import) / list what is known (lg_tune_export) / forget everything, first sights included (lg_tune_clear)
void remember(const Key &key, int choice);
size_t snapshot(Key *keys, int *choices, size_t capacity);
void clear();

} // namespace tune
} // namespace lg
