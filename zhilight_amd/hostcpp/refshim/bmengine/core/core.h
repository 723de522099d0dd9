#pragma once
#include "cublas_v2.h"
#include "bm_hip.h"
#include "bm_layer.h"
#include "bm_engine.h"
#include "bmengine/core/exception.h"
