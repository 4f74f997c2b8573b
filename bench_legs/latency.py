"""bench leg: single-call latency of the drop-in entry points"""
import ctypes
import time


from .common import *  # noqa: F401,F403


def run(c):
    eng, H, others, do_cpu = c.eng, c.H, c.others, c.do_cpu
    lat = {}

    def us_per_call(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6
    blk = ctypes.create_string_buffer(192)
    st_ctr = ctypes.create_string_buffer(eng.lib.beltCTR_keep())
    eng.lib.beltCTRStart(st_ctr, H[128:160], ctypes.c_size_t(32), H[192:208])
    b16 = ctypes.create_string_buffer(16)
    b64k = ctypes.create_string_buffer(1 << 16)
    import goldenlib
    Gk = goldenlib.Golden()
    h0, s0, p0 = Gk.bign_base[0]
    d0 = bytes(range(1, 33))
    sg0 = ctypes.create_string_buffer(48)

    def measure(fast):
        lat = {}
        lat["bashF (192 B)"] = us_per_call(lambda: eng.lib.bashF(blk, None), 20000 if fast else 200)
        lat["beltCTRStepE (16 B)"] = us_per_call(lambda: eng.lib.beltCTRStepE(b16, ctypes.c_size_t(16), st_ctr), 20000 if fast else 200)
        lat["beltCTRStepE (64 KiB)"] = us_per_call(lambda: eng.lib.beltCTRStepE(b64k, ctypes.c_size_t(1 << 16), st_ctr), 100)
        lat["bign128Verify"] = us_per_call(lambda: eng.lib.bign128Verify(h0, s0, p0), 1000 if fast else 50)
        lat["bign128PubkeyVal"] = us_per_call(lambda: eng.lib.bign128PubkeyVal(p0), 20000 if fast else 100)
        lat["bign128Sign2"] = us_per_call(lambda: eng.lib.bign128Sign2(sg0, h0, d0, None, ctypes.c_size_t(0)), 50)
        lat["beltHash (1 KiB)"] = us_per_call(lambda: eng.lib.beltHash(ctypes.create_string_buffer(32), bytes(1024), ctypes.c_size_t(1024)), 2000 if fast else 100)
        return lat
    lat = measure(True)                                        # the default: small calls on the host path, by size
    eng.lib.bee2hip_path_policy(1)                             # as BEE2HIP_FORCE=gpu: every primitive in a kernel (rounds 1-2)
    lat_gpu = measure(False)
    eng.lib.bee2hip_path_policy(0)
    entry = {"unit": "us per call", "dropin": lat, "dropin_forced_gpu": lat_gpu,
             "note": "dropin = the library as a caller gets it: single primitives, block-parallel modes under 8 KiB per call, "
                     "one-message serial chains and ONE signature verification / public-key validation run on the host path "
                     "(bee2_amd/csrc/host_small.hpp, host_bign.hpp; nothing with a private key does), everything else is H2D + "
                     "launch(es) + D2H on the NULL stream; dropin_forced_gpu = BEE2HIP_FORCE=gpu (every primitive in a kernel); "
                     "the batch entry points are the fast path (INTEGRATION.md gives the crossover sizes)"}
    if do_cpu:
        import refgen
        if refgen.have_ref():
            ref = ctypes.CDLL(refgen.REF_SO)
            cpu = {}
            cpu["bashF (192 B)"] = us_per_call(lambda: ref.bashF(blk, None), 20000)
            rst = ctypes.create_string_buffer(ref.beltCTR_keep())
            ref.beltCTRStart(rst, H[128:160], ctypes.c_size_t(32), H[192:208])
            cpu["beltCTRStepE (16 B)"] = us_per_call(lambda: ref.beltCTRStepE(b16, ctypes.c_size_t(16), rst), 20000)
            cpu["beltCTRStepE (64 KiB)"] = us_per_call(lambda: ref.beltCTRStepE(b64k, ctypes.c_size_t(1 << 16), rst), 500)
            cpu["bign128Verify"] = us_per_call(lambda: ref.bign128Verify(h0, s0, p0), 300)
            cpu["bign128PubkeyVal"] = us_per_call(lambda: ref.bign128PubkeyVal(p0), 5000)
            cpu["bign128Sign2"] = us_per_call(lambda: ref.bign128Sign2(sg0, h0, d0, None, ctypes.c_size_t(0)), 300)
            cpu["beltHash (1 KiB)"] = us_per_call(lambda: ref.beltHash(ctypes.create_string_buffer(32), bytes(1024), ctypes.c_size_t(1024)), 2000)
            entry["cpu_reference"] = cpu
    others["single_call_latency_us"] = entry

