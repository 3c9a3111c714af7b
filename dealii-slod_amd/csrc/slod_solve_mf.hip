// k_solve_mf: placeholder until the MFMA-factorised kernel lands (never chosen: 0 tiles).
#include "slod_device.h"
int        slod_solve_mf_tiles(int, int) { return 0; }
size_t     slod_solve_mf_lds_bytes(int, int, int) { return ~(size_t)0; }
hipError_t slod_launch_solve_mf(int, const SlodKernelArgs &, int, size_t, hipStream_t) { return hipErrorInvalidValue; }
