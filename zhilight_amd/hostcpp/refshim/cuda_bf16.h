#pragma once
#include <hip/hip_bf16.h>
