#pragma once
#include "cublas_v2.h"
