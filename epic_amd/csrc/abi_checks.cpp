// abi_checks.cpp -- compile-time proof that the struct layout is the reference's
// (libepic/include/epic/harmonic/harmonic.h:44-64; ctypes twin libepic/python/epic/epic_harmonic.py:45-57).
#include <cstddef>

#include "../../include/epic/epic_abi.h"

static_assert(sizeof(epic::Harmonic) == 80, "Harmonic must be 80 bytes");
static_assert(offsetof(epic::Harmonic, n) == 0, "n");
static_assert(offsetof(epic::Harmonic, m) == 8, "m");
static_assert(offsetof(epic::Harmonic, u) == 16, "u");
static_assert(offsetof(epic::Harmonic, locked) == 24, "locked");
static_assert(offsetof(epic::Harmonic, epsilon) == 32, "epsilon");
static_assert(offsetof(epic::Harmonic, delta) == 36, "delta");
static_assert(offsetof(epic::Harmonic, numIterationsToStaggerCheck) == 40, "numIterationsToStaggerCheck");
static_assert(offsetof(epic::Harmonic, currentIteration) == 44, "currentIteration");
static_assert(offsetof(epic::Harmonic, d_m) == 48, "d_m");
static_assert(offsetof(epic::Harmonic, d_u) == 56, "d_u");
static_assert(offsetof(epic::Harmonic, d_locked) == 64, "d_locked");
static_assert(offsetof(epic::Harmonic, d_delta) == 72, "d_delta");
