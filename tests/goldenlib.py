"""Readers for the committed fixtures under tests/golden/ (written by tools/make_golden.py from
outputs of the reference).  Pure data: no oracle, no ctypes -- bench.py's GPU legs and the tools may
import this; anything that needs the CPU oracle imports orclib instead."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


class Golden:
    """committed fixtures written by tools/make_golden.py (outputs of the reference)"""

    def __init__(self):
        with open(os.path.join(GOLD, "stb_kat.json")) as f:
            self.kat = json.load(f)
        with open(os.path.join(GOLD, "belt_bash_random.json")) as f:
            self.belt_bash = json.load(f)
        with open(os.path.join(GOLD, "bign_edge.json")) as f:
            self.bign_edge = json.load(f)
        with open(os.path.join(GOLD, "bash256_1MiB.json")) as f:
            self.big = json.load(f)
        raw = open(os.path.join(GOLD, "bashf_random.bin"), "rb").read()
        self.bashf_in, self.bashf_out = raw[: len(raw) // 2], raw[len(raw) // 2:]
        raw = open(os.path.join(GOLD, "bign_base.bin"), "rb").read()
        self.bign_base = [(raw[i:i + 32], raw[i + 32:i + 80], raw[i + 80:i + 144])
                          for i in range(0, len(raw), 144)]
        self.H = bytes.fromhex(self.kat["beltH"])
        with open(os.path.join(GOLD, "bign_big_curves.json")) as f:
            self.bign_big = json.load(f)
        with open(os.path.join(GOLD, "bign_pubkey_val.json")) as f:
            self.bign_pubkey_val = json.load(f)
        with open(os.path.join(GOLD, "bign_oid_lengths.json")) as f:
            self.bign_oid_lengths = json.load(f)
        with open(os.path.join(GOLD, "bign_oid_long.json")) as f:
            self.bign_oid_long = json.load(f)
        with open(os.path.join(GOLD, "sigvfy_pipeline.json")) as f:
            self.sigvfy_pipeline = json.load(f)
        with open(os.path.join(GOLD, "belt_bde_random.json")) as f:
            self.belt_bde = json.load(f)
        with open(os.path.join(GOLD, "belt_sde_random.json")) as f:
            self.belt_sde = json.load(f)
        with open(os.path.join(GOLD, "belt_dwp.json")) as f:
            self.belt_dwp = json.load(f)
        with open(os.path.join(GOLD, "belt_che.json")) as f:
            self.belt_che = json.load(f)
        with open(os.path.join(GOLD, "bign_sign.json")) as f:
            self.bign_sign = json.load(f)

    def bign_base_arrays(self):
        hs = b"".join(t[0] for t in self.bign_base)
        ss = b"".join(t[1] for t in self.bign_base)
        ps = b"".join(t[2] for t in self.bign_base)
        return hs, ss, ps
