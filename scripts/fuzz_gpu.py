#!/usr/bin/env python3
"""dev tool: time-bounded differential fuzz of the stage-1 device entry point against the CPU oracle (test
infrastructure, allowed here: scripts/ never ship).  Every case: random size (1 B .. 48 MiB, biased to tile and span
boundaries), a random mixture of generators (uniform random bytes over the special alphabet, CSV-like rows, blocks that
make the speculative scatter guess right or wrong with either entering state, dense runs), random misalignment,
base offset, entering state and tape capacity.  The whole tape, the count, the leaving state and the two hypothesis
counts must equal the oracle's.  Three more modes: a batch of buffers per launch, the host-buffer entry point, the dialect
extension with random delimiter / quote / escape bytes, and the UTF-8 validation pass against CPython's decoder.  usage: fuzz_gpu.py [seconds] [seed] [only this mode]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
oracle = graft.load_oracle()
ALPHABET = np.frombuffer(b',"\n\ra \\\x00\xff', dtype=np.uint8)


def gen_random(rng, n):
    w = np.ones(ALPHABET.size)
    w[1] = rng.choice([0.0, 0.01, 0.1, 1.0, 4.0])
    w[4] = rng.choice([1.0, 20.0, 200.0])
    return ALPHABET[rng.choice(ALPHABET.size, size=n, p=w / w.sum())].astype(np.uint8)


def gen_rows(rng, n):
    width = int(rng.integers(1, 60))
    cols = int(rng.integers(1, 40))
    field = b"x" * width
    q = rng.random() < 0.5
    row = b",".join((b'"' + field[:-2] + b',"' if (q and i % 3 == 1 and width > 3) else field) for i in range(cols))
    row += b"\r\n" if rng.random() < 0.3 else b"\n"
    return np.frombuffer((row * (n // len(row) + 1))[:n], dtype=np.uint8).copy()


def gen_quoted_body(rng, n):
    """one long quoted field full of separators: the speculation's guess is wrong for whoever enters it outside"""
    b = np.full(n, ord("y"), dtype=np.uint8)
    step = int(rng.integers(2, 40))
    b[::step] = rng.choice([0x2C, 0x0A, 0x0D])
    if n > 2:
        b[int(rng.integers(0, min(n, 64)))] = 0x22
        b[n - 1 - int(rng.integers(0, min(n - 1, 64)))] = 0x22
    return b


def gen_dense(rng, n):
    return np.full(n, rng.choice([0x2C, 0x0A, 0x22, 0x61]), dtype=np.uint8)


GENS = [gen_random, gen_rows, gen_quoted_body, gen_dense]


def make_case(rng):
    T = pkg.tile_bytes()
    kind = rng.integers(0, 4)
    if kind == 0:
        n = int(rng.integers(1, 4096))
    elif kind == 1:
        n = int(rng.integers(1, 20)) * (T // 8) + int(rng.integers(-70, 70))
    elif kind == 2:
        n = int(rng.integers(1, 12)) * T + int(rng.integers(-5000, 5000))
    else:
        n = int(rng.integers(1, 48 << 20))
    n = max(n, 1)
    parts, left = [], n
    while left > 0:
        m = left if rng.random() < 0.3 else int(rng.integers(1, left + 1))
        if rng.random() < 0.5:
            m = min(left, max(1, (m // T) * T + int(rng.integers(0, 3)) * (T // 8)))
        parts.append(GENS[int(rng.integers(0, len(GENS)))](rng, m))
        left -= m
    return np.concatenate(parts)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    only_mode = int(sys.argv[3]) if len(sys.argv) > 3 else None   # 0 batch, 1 host buffer, 2 dialect, 3 UTF-8, 4.. device entry
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(0)
    t_end = time.time() + seconds
    cases = bytes_total = 0
    bad = []
    t_say = time.time() + 60
    while time.time() < t_end and not bad:
        if time.time() > t_say:   # a line a minute: a silent GPU command is taken for hung after seven
            print(f"# {cases} cases", file=sys.stderr, flush=True)
            t_say = time.time() + 60
        mode = int(rng.integers(0, 9)) if only_mode is None else only_mode
        if os.environ.get("FUZZ_TRACE"):   # which case is running (a hang is then the last line), and a watchdog per case
            import faulthandler
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(int(os.environ["FUZZ_TRACE"]), exit=True)
            print(f"case {cases} mode {mode}", file=sys.stderr, flush=True)
        if mode == 8:
            # MANY host files in one call (round 5: csvsimd_stage1_index_batch): every file's tape must be its own
            k = int(rng.integers(1, 400))
            pool = np.concatenate([make_case(rng), make_case(rng)])       # the files are slices of two generated cases
            files = []
            for _ in range(k):
                kind = rng.random()
                size = int(rng.integers(0, 64)) if kind < 0.1 else int(rng.integers(0, 40 << 10)) if kind < 0.95 else int(rng.integers(0, 3 << 20))
                size = min(size, pool.size)
                at = int(rng.integers(0, pool.size - size + 1))
                files.append(pool[at: at + size].copy())
            got = ctx.read_many(files)
            for i, (f, g) in enumerate(zip(files, got)):
                if not np.array_equal(g, oracle.scalar_read(f)):
                    bad.append({"case": cases, "mode": "host batch", "item": i, "n": int(f.size)})
                    break
            cases += 1
            bytes_total += sum(int(f.size) for f in files)
            continue
        if mode == 0:
            # a BATCH of buffers in one launch: every record and every tape must be the buffer's own
            k = int(rng.integers(1, 12))
            bufs = [make_case(rng)[: int(rng.integers(0, 6 << 20))] for _ in range(k)]
            states = [int(rng.integers(0, 2)) for _ in range(k)]
            bases = [int(rng.integers(0, 1 << 40)) for _ in range(k)]
            miss = [int(rng.integers(0, 128)) for _ in range(k)]
            dbufs, dtapes, items = [], [], []
            for b, st, ba, mi in zip(bufs, states, bases, miss):
                t = torch.full((b.size + 256,), 0x2C, dtype=torch.uint8, device="cuda:0")
                if b.size:
                    t[mi: mi + b.size] = torch.from_numpy(b)
                tp = torch.full((b.size + 9,), -1, dtype=torch.int64, device="cuda:0")
                dbufs.append(t); dtapes.append(tp)
                items.append((t.data_ptr() + mi, b.size, ba, tp.data_ptr(), b.size + 1, st))
            dres = torch.zeros((k, 8), dtype=torch.int64, device="cuda:0")
            ctx.stage1_index_batch_device_async(items, dres.data_ptr())
            torch.cuda.synchronize()
            for i, b in enumerate(bufs):
                want, q = oracle.scalar_index(b, base_off=bases[i], in_quote_in=states[i])
                r = pkg.ShardResult.from_buffer_copy(dres[i].cpu().numpy().tobytes())
                got = dtapes[i][: want.size].cpu().numpy().view(np.uint64)
                if not (r.count == want.size and r.in_quote_out == q and r.error == 0 and np.array_equal(got, want)
                        and bool((dtapes[i][want.size:] == -1).all())):
                    bad.append({"case": cases, "mode": "batch", "item": i, "n": int(b.size), "count": int(r.count), "want": int(want.size)})
            cases += 1
            bytes_total += sum(int(b.size) for b in bufs)
            continue
        if mode == 1:
            # the host-buffer entry point: chunks chained on the device, tape returned as 32-bit offsets, dense retries
            d = make_case(rng)
            want = oracle.scalar_read(d)
            got = ctx.read(d)
            if not (got.size == want.size and np.array_equal(got, want)):
                bad.append({"case": cases, "mode": "host", "n": int(d.size), "count": int(got.size), "want": int(want.size)})
                np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_host_{seed}_{cases}.npy"), d)
            cases += 1
            bytes_total += int(d.size)
            continue
        if mode == 2:
            # the dialect extension: ANY delimiter / quote / escape bytes (every triple takes the hashed or the compare
            # classification, whichever the host search finds), every byte value in the data, the special ones often
            while True:
                dl, qu, es = int(rng.integers(1, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 256))
                qu = 0 if rng.random() < 0.15 else qu
                es = 0 if rng.random() < 0.3 else es
                if dl in (10, 13) or qu in (dl, 10, 13) or es in (dl, 10, 13) or (es and es == qu):
                    continue
                break
            n = int(rng.choice([int(rng.integers(1, 5000)), int(rng.integers(1, 3 * pkg.tile_bytes()))]))
            d = rng.integers(0, 256, size=n, dtype=np.uint8)
            special = np.array([b for b in (dl, qu, es, 10, 13) if b], dtype=np.uint8)
            near = np.array([int(b) ^ (1 << k) for b in special for k in range(8)], dtype=np.uint8)   # one bit off
            u = rng.random(n)
            p_sp = rng.choice([0.02, 0.15, 0.6])
            d[u < p_sp] = special[rng.integers(0, special.size, size=int((u < p_sp).sum()))]
            d[u > 0.9] = near[rng.integers(0, near.size, size=int((u > 0.9).sum()))]
            dia = pkg.Dialect(dl, qu or None, es or None, escape_in=int(rng.integers(0, 2)) if es else 0)
            inq, base, mis = int(rng.integers(0, 2)), int(rng.integers(0, 1 << 40)), int(rng.integers(0, 128))
            want, q, e = oracle.dialect_index(d, dl, qu, es, base_off=base, in_quote_in=inq, escape_in=dia.escape_in)
            dbuf = torch.full((n + 256,), es or dl, dtype=torch.uint8, device="cuda:0")
            dbuf[mis: mis + n] = torch.from_numpy(d)
            dtape = torch.full((n + 9,), -1, dtype=torch.int64, device="cuda:0")
            dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            ctx.stage1_index_device_dialect_async(dia, dbuf.data_ptr() + mis, n, base, inq, dtape.data_ptr(), n + 1,
                                                  dres.data_ptr())
            torch.cuda.synchronize()
            r = pkg.ShardResult.from_buffer_copy(dres.cpu().numpy().tobytes())
            got = dtape[: want.size].cpu().numpy().view(np.uint64)
            if not (r.error == 0 and r.count == want.size and r.written == r.count and np.array_equal(got, want)
                    and bool((dtape[want.size:] == -1).all()) and r.in_quote_out == q and (not es or r.escape_out == e)):
                bad.append({"case": cases, "mode": "dialect", "dialect": [dl, qu, es, int(dia.escape_in)], "n": n,
                            "count": int(r.count), "want": int(want.size)})
                np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_dialect_{seed}_{cases}.npy"), d)
            cases += 1
            bytes_total += n
            continue
        if mode == 3:
            # UTF-8 validation (extension): text in one to four byte sequences, a few random corruptions, random cut
            cps = np.concatenate([rng.integers(0x20, 0x7F, size=int(rng.integers(0, 30000))),
                                  rng.integers(0xA0, 0x7FF, size=int(rng.integers(0, 20000))),
                                  rng.integers(0x800, 0xD7FF, size=int(rng.integers(0, 20000))),
                                  rng.integers(0xE000, 0xFFFF, size=int(rng.integers(0, 3000))),
                                  rng.integers(0x10000, 0x10FFFF, size=int(rng.integers(0, 20000)))])
            rng.shuffle(cps)
            d = np.frombuffer("".join(map(chr, cps.tolist())).encode("utf-8"), dtype=np.uint8).copy()
            if d.size == 0:
                continue
            d = d[int(rng.integers(0, min(4, d.size))): d.size - int(rng.integers(0, min(4, d.size)))]
            for p_ in rng.integers(0, max(d.size, 1), size=int(rng.integers(0, 4))):
                if d.size:
                    d[p_] = rng.integers(0, 256)
            n, mis = d.size, int(rng.integers(0, 128))
            dbuf = torch.full((n + 256,), int(rng.choice([0xFF, 0x80, 0xE2])), dtype=torch.uint8, device="cuda:0")
            if n:
                dbuf[mis: mis + n] = torch.from_numpy(d)
            got = ctx.utf8_validate_device(dbuf.data_ptr() + mis, n)
            want = oracle.utf8_first_invalid(d)
            if got != want:
                bad.append({"case": cases, "mode": "utf8", "n": n, "got": got, "want": want})
                np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_utf8_{seed}_{cases}.npy"), d)
            cases += 1
            bytes_total += n
            continue
        d = make_case(rng)
        n = d.size
        mis = int(rng.integers(0, 128))
        inq = int(rng.integers(0, 3))   # 2 = CSVSIMD_ENTER_GUESS: the kernel chooses; the record says what it used
        base = int(rng.integers(0, 1 << 40))
        if inq == 2:
            _, a0, b0 = oracle.shard_descriptor(d[: 8 * pkg.tile_bytes()])
            used = int(b0 > a0)             # the documented rule: the state under which the first EIGHT tiles have more entries
        else:
            used = inq
        want, q = oracle.scalar_index(d, base_off=base, in_quote_in=used)
        cap_kind = rng.integers(0, 3)
        cap = want.size + 3 if cap_kind == 0 else (n + 1 if cap_kind == 1 else int(rng.integers(0, want.size + 1)))
        dbuf = torch.full((n + 256,), 0x2C, dtype=torch.uint8, device="cuda:0")
        dbuf[mis: mis + n] = torch.from_numpy(d)
        dtape = torch.full((cap + 8,), -1, dtype=torch.int64, device="cuda:0")
        # either instantiation (round 4: the dense one has another geometry and another emit path, the same results)
        ctx.hint_density(1, 2) if rng.random() < 0.5 else ctx.hint_density(1, 1000)
        # any grid (round 5: a launch, CSVSIMD_ENTER_GUESS included, needs no particular number of resident workgroups)
        ctx.limit_workgroups(int(rng.choice([0, 0, 0, 1, 2, 3, 7, 33, 200])))
        r = ctx.stage1_index_device(dbuf.data_ptr() + mis, n, base, inq, dtape.data_ptr(), cap, allow_overflow=True)
        ctx.limit_workgroups(0)
        torch.cuda.synchronize()
        k = min(r.count, cap)
        got = dtape[:k].cpu().numpy().view(np.uint64)
        p, c0, c1 = oracle.shard_descriptor(d)
        ok = (r.count == want.size and r.in_quote_out == q and r.written == k and np.array_equal(got, want[:k])
              and bool((dtape[k:] == -1).all()) and r.error == 0 and r.in_quote_in_used == used
              and (r.quote_parity, r.count_enter_outside, r.count_enter_inside) == (p, c0, c1))
        if not ok:
            bad.append({"case": cases, "n": n, "mis": mis, "inq": inq, "cap": cap, "count": int(r.count),
                        "want": int(want.size)})
            np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{cases}.npy"), d)
        cases += 1
        bytes_total += n
        del dbuf, dtape
    print(json.dumps({"seconds": seconds, "seed": seed, "cases": cases, "GiB": round(bytes_total / 2**30, 2), "bad": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
