// compat/evaluator.h -- see compat/marching.h.
#pragma once
#include "marching.h"
