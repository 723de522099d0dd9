// refshim: bmengine/c10d/c10d.h -> the collective wrappers of the HIP shim
#pragma once
#include "bm_c10d.h"
