#pragma once
#include "bm_functions.h"
#include "bmengine/functions/element.h"
#include "bmengine/functions/init.h"
#include "bmengine/functions/scatter.h"
