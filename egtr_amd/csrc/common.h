// Shared helpers for the libegtr_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/egtr_hip.h"

// Records hipGetLastError() for egtr_last_hip_error() and maps it to an EGTR_* status.
int egtr_check_launch();

// Dynamic LDS above 64 KiB has to be requested with hipFuncSetAttribute -- per DEVICE (the attribute belongs to the device's
// copy of the code object).  `done` is the kernel's own bit mask of device ordinals already served (a race between two
// host threads only sets the same value twice).  Returns EGTR_OK or the launch status.
int egtr_raise_dynamic_lds(const void* kernel, int bytes, unsigned long long* done);

// ReLU as torch computes it: relu(NaN) = NaN (fmaxf / v_max_f32 would return 0 and hide a diverged activation).
__device__ __forceinline__ float egtr_relu(float x) { return x < 0.f ? 0.f : x; }
