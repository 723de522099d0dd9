#pragma once
#include "bm_functions.h"
