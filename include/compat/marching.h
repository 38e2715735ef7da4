// compat/marching.h -- lets a caller written against the reference's Source/marching.h compile unchanged: put
// include/compat first on the include path and link libmc_hip.so.  The classes are the GPU-backed ones of
// include/mc_marching.hpp (same names, same methods; see the table at the top of that file).
#pragma once
#include "../mc_marching.hpp"
using mc_amd::CalculateNormal;  // Source/normal.h:3
using mc_amd::Evaluator;        // Source/evaluator.h:24 (the reference's marching.h includes evaluator.h too)
using mc_amd::Marching;         // Source/marching.h:72
using mc_amd::Poly_Data;        // Source/marching.h:26
using mc_amd::Step_Data;        // Source/marching.h:15
