"""bee2_amd -- MI355X (gfx950) batch engine for bee2's hot paths (bashF, belt CTR/MAC,
bign verify).  Thin ctypes view of the C ABI in include/bee2hip.h; the compute is
hand-written HIP in bee2_amd/csrc.  There is no CPU fallback: importing works
anywhere, but any primitive call needs libbee2hip.so and a GPU."""
from .engine import (  # noqa: F401
    Engine,
    EngineError,
    ERR_BAD_INPUT,
    ERR_BAD_OID,
    ERR_BAD_PARAMS,
    ERR_BAD_PUBKEY,
    ERR_BAD_SIG,
    ERR_OK,
    EXP_LIB_PATH,
    LIB_PATH,
    build,
    lib_exports,
    load,
    load_experiments,
)
