#pragma once
#include "cublas_v2.h"
#include "bm_hip.h"
// exception.h:13-22: the branch hint layer code uses directly (feedforward.cpp:571)
static inline bool bm_unlikely(bool x) { return __builtin_expect(x, 0); }
