#pragma once
#include "bm_layer.h"
