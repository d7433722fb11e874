// Shared helpers for the libegtr_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/egtr_hip.h"

// Records hipGetLastError() for egtr_last_hip_error() and maps it to an EGTR_* status.
int egtr_check_launch();
