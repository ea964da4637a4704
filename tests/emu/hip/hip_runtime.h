// Stand-in for <hip/hip_runtime.h> when the kernel sources are compiled for the HOST by
// tests/emu/build_emu.py.  TEST INFRASTRUCTURE ONLY (see tests/emu/hip_emu.h).
#pragma once
#include "../hip_emu.h"
