#pragma once
#include "bm_layer.h"
#include "bm_engine.h"
