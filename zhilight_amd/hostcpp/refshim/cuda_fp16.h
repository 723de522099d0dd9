#pragma once
#include <hip/hip_fp16.h>
