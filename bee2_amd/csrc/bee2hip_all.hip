// bee2hip_all.hip -- single translation unit of libbee2hip.so.
// The kernels share __constant__ data (the belt S-box, curve tables); building them
// as one TU avoids relocatable device code.  Build: see bee2_amd/csrc/Makefile.
#include "bash_kernels.hip"
#include "belt_kernels.hip"
#include "bign_kernels.hip"
#include "bign_sign_kernels.hip"
#include "bign_generic_kernels.hip"
#include "mixed_kernels.hip"
#include "capi.hip"
#include "multi.hip"
