// bee2hip_tu_belt.hip -- translation unit 1 of 2 of libbee2hip.so: bash-f, belt, the fused / ragged hashing kernels and
// the C ABI.  The two units share no device symbol (each holds its own copy of the belt S-box, belt_dev.hpp), so they
// compile side by side without relocatable device code.  Build: bee2_amd/csrc/Makefile.
#include "bash_kernels.hip"
#include "belt_kernels.hip"
#include "mixed_kernels.hip"
#include "capi.hip"
#include "multi.hip"
