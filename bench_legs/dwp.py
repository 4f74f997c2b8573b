"""bench leg: belt-dwp / belt-che"""
import ctypes
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, K, N, H, others, cores, do_cpu = c.dist, c.eng, c.K, c.N, c.H, c.others, c.cores, c.do_cpu
    nbytes = 4 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < nbytes + (1 << 30):
        nbytes = (int(free * 0.5) // (1 << 20)) << 20
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    fill_seeded(buf, 0xBE17 + dist.rank)
    dkw, dc0, dr, dt0 = eng.beltDWPStart(H[128:160], H[192:208])
    tout = torch.zeros(16, dtype=torch.uint8, device="cuda")
    kd = max(3, min(K, 10))

    def wrap_dev():                       # beltDWPWrap on resident data: encrypt in place, authenticate the ciphertext
        eng.beltCTR_blocks_dev(buf, dkw, dc0, 0)
        eng.beltDWP_absorb_dev(buf, nbytes, dr, dt0, tout)
    el = timed(dist, kd, 1, wrap_dev)
    el_mac = timed(dist, kd, 1, lambda: eng.beltDWP_absorb_dev(buf, nbytes, dr, dt0, tout))
    ckw, cs0, ct0 = eng.beltCHEStart(H[128:160], H[192:208])

    def che_wrap_dev():                   # beltCHEWrap on resident data (r = s0 for belt-che)
        eng.beltCHE_blocks_dev(buf, buf, ckw, cs0, 0)
        eng.beltDWP_absorb_dev(buf, nbytes, cs0, ct0, tout)
    el_che = timed(dist, kd, 1, che_wrap_dev)
    entry = {"metric": "belt-dwp wrap GiB/s", "unit": "GiB/s", "value": N * nbytes * kd / el / 2 ** 30, "steps": kd,
             "ms_per_step": el / kd * 1e3, "mac_only": N * nbytes * kd / el_mac / 2 ** 30,
             "che_wrap": N * nbytes * kd / el_che / 2 ** 30,
             "config": {"workload": f"{nbytes / 2**30:.0f} GiB message per GPU, device resident: beltCTR_blocks_dev + "
                                    "beltDWP_absorb_dev over the ciphertext (each rank its own message; SURVEY 8f-2)"}}
    if do_cpu:
        import refgen
        if refgen.have_ref():
            from concurrent.futures import ThreadPoolExecutor
            ref = ctypes.CDLL(refgen.REF_SO)
            nthr = min(cores, 64)
            per = 4 << 20                                      # 4 MiB message per thread
            hb = np.ones(nthr * per, dtype=np.uint8)
            macs = np.zeros((nthr, 8), dtype=np.uint8)
            key, iv = bytes(H[128:160]), bytes(H[192:208])

            def work(t):
                p = ctypes.c_void_p(hb.ctypes.data + t * per)
                ref.beltDWPWrap(p, ctypes.c_void_p(macs[t].ctypes.data), p, ctypes.c_size_t(per), None,
                                ctypes.c_size_t(0), key, ctypes.c_size_t(32), iv)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as ex:
                list(ex.map(work, range(nthr)))
            dt = time.perf_counter() - t0
            # same message through the GPU drop-in: the tags must agree
            one = np.ones(per, dtype=np.uint8)
            code, _, gmac = eng.dwp_wrap(one.tobytes(), b"", key, iv)
            entry["cpu_baseline"] = {"kind": "reference", "cores": nthr, "unit": "GiB/s", "value": nthr * per / dt / 2 ** 30,
                                     "sample": f"{nthr} threads, one 4 MiB beltDWPWrap each",
                                     "mac_equal": bool(code == 0 and gmac == macs[0].tobytes())}
            cm = (ctypes.c_ubyte * 8)()
            cb = np.ones(per, dtype=np.uint8)
            t0 = time.perf_counter()
            ref.beltCHEWrap(ctypes.c_void_p(cb.ctypes.data), cm, ctypes.c_void_p(cb.ctypes.data), ctypes.c_size_t(per),
                            None, ctypes.c_size_t(0), key, ctypes.c_size_t(32), iv)
            dt1 = time.perf_counter() - t0
            code, _, gmac = eng.dwp_wrap(one.tobytes(), b"", key, iv, "CHE")
            entry["cpu_baseline"]["che_wrap_single_thread"] = per / dt1 / 2 ** 30
            entry["cpu_baseline"]["che_mac_equal"] = bool(code == 0 and gmac == bytes(cm))
    others["belt_dwp"] = entry
    del buf, tout

