#!/usr/bin/env python3
"""dev tool: time-bounded differential fuzz of the UTF-8 validation pass against CPython's decoder.  Texts are assembled from
pieces of several scripts (so that 1-KiB wave chunks fall into every tier of the rules: ASCII, 2-byte and 3-byte text without
special leads, E0 / ED leads, F0 leads) with runs of one script long enough to fill whole chunks, then corrupted at random
places with random bytes, truncated, and validated at random misalignments.  usage: fuzz_utf8.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = pkg.Context(0)
PIECES = ["plain,ascii,row\n", "città,naïve café,Ünïcödé\n", "Съешь же ещё этих мягких булок\n", "漢字仮名交じり文東京都",
          "ภาษาไทยเป็นภาษา", "한국어는 한반도에서 힘", "\U0001F680\U0001F600 ok ", "ࠀ￿퟿", "\U00010000\U0010ffff",
          "\x7f\x00a"]
ENC = [p.encode() for p in PIECES]
def first_invalid(b: bytes):
    try:
        b.decode("utf-8")
        return None
    except UnicodeDecodeError as e:
        return e.start
cases = bad = 0
t_end = time.time() + budget
while time.time() < t_end:
    parts = []
    total = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20000), rng.integers(20000, 400000)]))
    size = 0
    while size < total:
        p = ENC[int(rng.integers(0, len(ENC)))]
        rep = int(rng.choice([1, 3, 40, 400]))   # long runs of one script fill whole 1-KiB chunks
        parts.append(p * rep)
        size += len(p) * rep
    data = bytearray(b"".join(parts)[:total + int(rng.integers(0, 4))])
    kind = int(rng.integers(0, 4))
    if kind >= 1:   # corruptions: random bytes at random places (kind 3: many)
        for _ in range(1 if kind == 1 else int(rng.integers(1, 4)) if kind == 2 else int(rng.integers(4, 40))):
            if data:
                data[int(rng.integers(0, len(data)))] = int(rng.choice([0x80, 0xBF, 0xC0, 0xC1, 0xE0, 0xED, 0xF0, 0xF4, 0xF5, 0xFF, int(rng.integers(0, 256))]))
    b = bytes(data)
    want = first_invalid(b)
    mis = int(rng.integers(0, 128))
    poison = int(rng.choice([0xFF, 0x80, 0x00]))
    dbuf = torch.full((len(b) + 256,), poison, dtype=torch.uint8, device="cuda:0")
    if b:
        dbuf[mis: mis + len(b)] = torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy())
    got = ctx.utf8_validate_device(dbuf.data_ptr() + mis, len(b))
    cases += 1
    if got != want:
        bad += 1
        print("MISMATCH", len(b), mis, poison, kind, got, want, b[max(0, (want or got or 0) - 8): (want or got or 0) + 8], flush=True)
        if bad > 5:
            break
print("fuzz_utf8: %d cases in %.0f s, seed %d: %s" % (cases, budget, seed, "clean" if bad == 0 else "%d MISMATCHES" % bad))
sys.exit(1 if bad else 0)
