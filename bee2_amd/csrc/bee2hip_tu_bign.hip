// bee2hip_tu_bign.hip -- translation unit 2 of 2 of libbee2hip.so: the bign kernels (verification, key generation,
// signing, non-standard parameter sets) and their launchers.  See bee2hip_tu_belt.hip.
#include "bign_kernels.hip"
#include "bign_sign_kernels.hip"
#include "bign_generic_kernels.hip"
