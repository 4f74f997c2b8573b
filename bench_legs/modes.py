"""bench leg: ECB / CBC / BDE / SDE bulk modes"""
import ctypes
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403


def run(c):
    dist, eng, K, N, H, kw, c0, others, cores, do_cpu = c.dist, c.eng, c.K, c.N, c.H, c.kw, c.c0, c.others, c.cores, c.do_cpu
    nbytes = 4 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < 2 * nbytes + (1 << 30):
        nbytes = (int(free * 0.3) // (1 << 20)) << 20
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    fill_seeded(src, 0xBE17 + dist.rank)
    dst = torch.empty_like(src)
    km = max(3, min(K, 10))
    entry = {"metric": "belt ECB/CBC/BDE/SDE bulk GiB/s", "unit": "GiB/s", "steps": km,
             "config": {"workload": f"{nbytes / 2**30:.0f} GiB of full blocks per GPU, one key (SURVEY 8f-1)"}}
    for name, mode in (("ecb_encr", 0), ("ecb_decr", 1), ("cbc_decr", 2)):
        el = timed(dist, km, 1, lambda: eng.beltModes_blocks_dev(mode, src, dst, kw, c0))
        entry[name] = N * nbytes * km / el / 2 ** 30
    # belt-bde: rank r owns blocks [r*nb, (r+1)*nb) of one stream (first_block), like CTR
    bkw, bs0 = eng.beltBDEStart(H[128:160], H[192:208])
    for name, decr in (("bde_encr", 0), ("bde_decr", 1)):
        el = timed(dist, km, 1, lambda: eng.beltBDE_blocks_dev(decr, src, dst, bkw, bs0,
                                                               first_block=dist.rank * (nbytes // 16)))
        entry[name] = N * nbytes * km / el / 2 ** 30
    # belt-sde: independent sectors, one lane each (a sector is a serial chain of 2 E per block)
    for sb in (512, 4096):
        ns = (1 << 20) if sb == 512 else (1 << 19)             # 512 MiB / 2 GiB of sectors: >= 8 wavefronts per SIMD
        ns = min(ns, nbytes // sb, dst.numel() // 16)
        sec = src[: ns * sb]
        sivs = dst[: 16 * ns]
        fill_seeded(sivs, 0x5DE + dist.rank)
        for name, decr in ((f"sde_encr_{sb}", 0), (f"sde_decr_{sb}", 1)):
            el = timed(dist, km, 1, lambda: eng.beltSDE_sectors_dev(decr, sec, sb, kw, sivs))
            entry[name] = N * ns * sb * km / el / 2 ** 30
    # CBC encryption is a serial chain per message: a batch of independent 512-byte messages, one lane each
    nm, mb = 1 << 20, 512
    cmsgs = src[: nm * mb]
    civs = dst[: 16 * nm]
    fill_seeded(civs, 0xCBC + dist.rank)
    el = timed(dist, km, 1, lambda: eng.beltCBCEncr_batch_dev(cmsgs, mb // 16, kw, civs))
    entry["cbc_encr_batch_512"] = N * nm * mb * km / el / 2 ** 30
    entry["value"] = entry["ecb_encr"]
    entry["ms_per_step"] = nbytes / 2 ** 30 / entry["ecb_encr"] * 1e3 * N
    if do_cpu:
        import orclib
        import refgen
        if refgen.have_ref():
            orc = orclib.load()
            ref = ctypes.CDLL(refgen.REF_SO)
            hb = np.zeros(64 << 20, dtype=np.uint8)
            ho = np.empty_like(hb)
            cpu = {"cores": cores, "kind": "reference", "unit": "GiB/s",
                   "sample": "64 MiB of full blocks, threads over disjoint slices (belt-sde: each slice one sector)"}
            for name, fn, iv in (("ecb_encr", "beltECBEncr", None), ("ecb_decr", "beltECBDecr", None),
                                 ("cbc_decr", "beltCBCDecr", H[192:208]), ("bde_encr", "beltBDEEncr", H[192:208]),
                                 ("bde_decr", "beltBDEDecr", H[192:208]), ("sde_encr", "beltSDEEncr", H[192:208]),
                                 ("sde_decr", "beltSDEDecr", H[192:208])):
                fp = ctypes.cast(getattr(ref, fn), ctypes.c_void_p)
                t0, reps = time.perf_counter(), 0
                while time.perf_counter() - t0 < 1.5:
                    orc.lib.orc_drive_ref_mode(fp, ctypes.c_void_p(hb.ctypes.data), ctypes.c_void_p(ho.ctypes.data),
                                               ctypes.c_size_t(hb.nbytes // 16), H[128:160], ctypes.c_size_t(32),
                                               iv, cores)
                    reps += 1
                cpu[name] = reps * hb.nbytes / 2 ** 30 / (time.perf_counter() - t0)
            cpu["value"] = cpu["ecb_encr"]
            entry["cpu_baseline"] = cpu
    others["belt_modes"] = entry
    del src, dst

