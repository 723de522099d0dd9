// refshim: bmengine/functions/all.h -- everything the reference's umbrella header pulls in (functions/all.h:1-13)
#pragma once
#include "bm_functions.h"
#include "bmengine/functions/element.h"
#include "bmengine/functions/init.h"
#include "bmengine/functions/scatter.h"
#include "bmengine/functions/softmax.h"
#include "bmengine/functions/topk.h"
