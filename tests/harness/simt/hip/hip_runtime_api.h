// see hip_runtime.h in this directory (SIMT shim, test infrastructure)
#pragma once
#include "hip_runtime.h"
