// capi.hip -- the C ABI of libbee2hip.so (declared in include/bee2hip.h).
//
// Host side of the engine: argument checks and state bookkeeping mirror bee2's
// C functions line for line in *behaviour* (same names, same state layouts, same
// error codes).  Every batch / _dev / _multi entry point and every bign operation evaluates its primitives in
// kernels; the bee2 drop-in symbols do so too, except for small single calls, which take the host path of
// host_small.hpp ("host path for small single calls" below says exactly when).
// Split by primitive in round 5 (one translation unit still: the files below are included in this order, nothing else changed):
//   staging.hpp     errors, per-device constants, scratch pool, staging, host-path policy, duplex pipeline
//   capi_base.hip   management, primitive batch entries, shared helpers
//   capi_bash.hip   bashF, bashHash*
//   capi_belt.hip   belt block / CTR / MAC / modes / AEAD / belt-hash
//   capi_bign.hip   bign verification, validation, key generation, signing
//   capi_mixed.hip  bash + belt-MAC per message, ragged hash batches, path policy
//   capi_exp.hip    experiment hooks (libbee2hip_exp.so only)
#include "staging.hpp"
#include "capi_base.hip"
#include "capi_bash.hip"
#include "capi_belt.hip"
#include "capi_bign.hip"
#include "capi_mixed.hip"
#include "capi_exp.hip"
