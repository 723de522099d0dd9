// refshim: private/allocator.h -> core::MemoryAllocator of the HIP shim (the arena base address is all layer code asks for)
#pragma once
#include "bm_hip.h"
