#pragma once
#include "bm_hip.h"
