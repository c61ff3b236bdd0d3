#!/usr/bin/env python3
"""dev tool: time-bounded differential fuzz of the columnar consumers (csvsimd_chunk_to_columns_device, then the frequency
count and the search on a column of the copy) against the scalar definitions in oracle/ (test infrastructure, allowed
here: scripts/ never ship).  Every case is a random RECTANGULAR file: 1..3000 columns, LF or CRLF, field lengths from 0 to
beyond the LDS window (rows longer than 16 KiB, more than 2048 tape entries per row), quoted fields holding separators
and line ends, a low-cardinality vocabulary mixed in; random chunking, field lists (with repeats), strides; every cell
(or a sample of 4000 on large cases) must be seek_field's text truncated and zero padded, every length exact, the count a
collections.Counter and the search Python's == / startswith / in; then the per-column consumers on the row-major file
itself (chunk spans, exact count over all chunks with values of any length, search).  usage: fuzz_columnar.py [seconds] [seed]"""
import json
import os
import sys
import time
from collections import Counter

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

pkg = graft.load_package()
oracle = graft.load_oracle()

VOCAB = [b"Oslo", b"", b"Bergen", b'"a,b"', b'"line\nbreak"', b"S\xc3\xa3o Paulo", b"x" * 15, b"x" * 16, b"x" * 17,
         b'"q""q"', b" ", b"0"]


def make_field(rng, kind):
    if kind == 0:
        return VOCAB[int(rng.integers(0, len(VOCAB)))]
    if kind == 1:
        return bytes(rng.integers(97, 123, size=int(rng.integers(0, 41)), dtype=np.uint8))
    if kind == 2:   # long: around the stride sizes, sometimes beyond the LDS window
        n = int(rng.choice([63, 64, 65, 255, 256, 257, 4095, 4097, 9000, 17000, 40000]))
        return bytes(rng.integers(97, 123, size=n, dtype=np.uint8))
    body = bytes(rng.choice(np.frombuffer(b'ab,\n\r ', dtype=np.uint8), size=int(rng.integers(0, 30))))
    return b'"' + body + b'"'


def make_file(rng):
    shape = int(rng.integers(0, 6))
    if shape == 0:
        cols, rows = int(rng.integers(2049, 3000)), int(rng.integers(1, 6))          # more entries than the window stages
    elif shape == 1:
        cols, rows = int(rng.integers(1, 5)), int(rng.integers(1, 40))
    else:
        cols, rows = int(rng.integers(1, 70)), int(rng.integers(1, 3000))
    rows = max(1, min(rows, 400_000 // cols))
    p_long = rng.choice([0.0, 0.0, 0.002, 0.02]) if shape != 0 else 0.0
    p_quoted = rng.choice([0.0, 0.05, 0.4])
    p_vocab = rng.choice([0.0, 0.3, 0.9])
    line_end = b"\r\n" if rng.random() < 0.4 else b"\n"
    out = [b",".join(b"h%d" % c for c in range(cols))]
    budget = 6 << 20
    for _ in range(rows):
        fs = []
        for _c in range(cols):
            u = rng.random()
            kind = 2 if u < p_long else (3 if u < p_long + p_quoted else (0 if rng.random() < p_vocab else 1))
            fs.append(make_field(rng, kind))
        row = b",".join(fs)
        budget -= len(row)
        out.append(row)
        if budget < 0:
            break
    if cols == 1 and out[1] == b"":
        out[1] = b"z"   # "h0\n\n...": Header::new (src/tape.rs:226-273) reads LF LF after the header as a CRLF file
    return line_end.join(out) + line_end


def one_case(rng, ctx):
    data = make_file(rng)
    one_case.last_data = data
    index = ctx.read(data)
    tape = pkg.Tape.from_index(np.frombuffer(data, dtype=np.uint8), index)
    F, crlf = tape.field_cnt, tape.new_line == "CRLF"
    nrec = tape.record_cnt - 1
    if nrec < 1:
        return 0, None
    mis = int(rng.integers(0, 16))
    dbuf = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda:0")
    dbuf[mis: mis + len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy())
    dindex = torch.from_numpy(index.view(np.int64).copy()).cuda()
    n_chunks = min(int(rng.integers(1, 9)), nrec)
    cells = 0
    for ch in tape.chunks(n_chunks):
        n = ch[3]
        if n == 0:
            continue
        k = int(rng.integers(0, 4))
        fields = None if (k == 0 and F <= 1024) else [int(x) for x in rng.integers(0, F, size=int(rng.integers(1, min(F, 12) + 1)))]
        flist = list(range(F)) if fields is None else fields
        stride = int(rng.choice([16, 32, 48, 64, 256, 1024]))
        if len(flist) * n * stride > (1 << 30):
            stride = 16
        cols = torch.full((len(flist), n, stride), 0xEE, dtype=torch.uint8, device="cuda:0")
        lens = torch.full((len(flist), n), -1, dtype=torch.int32, device="cuda:0")
        got_n = pkg.chunk_to_columns_device(ctx, dbuf.data_ptr() + mis, len(data), dindex.data_ptr(), index.size, F,
                                            tape.new_line, ch, fields, cols.data_ptr(), stride, lens.data_ptr())
        torch.cuda.synchronize()
        if got_n != n:
            return cells, {"what": "record count", "got": got_n, "want": n}
        h, hl = cols.cpu().numpy(), lens.cpu().numpy()
        recs = list(oracle.chunk_record_ids(ch, F, crlf))
        total = len(flist) * n
        if total <= 4000:
            pairs = [(c, i) for c in range(len(flist)) for i in range(n)]
        else:
            pairs = [(int(c), int(i)) for c, i in zip(rng.integers(0, len(flist), 4000), rng.integers(0, n, 4000))]
            pairs += [(c, i) for c in range(min(len(flist), 8)) for i in (0, n - 1)]
        for c, i in pairs:
            text = oracle.seek_field(data, index, F, crlf, recs[i], flist[c])
            want = text[:stride] + b"\0" * (stride - min(len(text), stride))
            if h[c, i].tobytes() != want or int(hl[c, i]) != len(text):
                return cells, {"what": "cell", "rec": recs[i], "field": flist[c], "stride": stride, "len": len(text),
                               "got_len": int(hl[c, i]), "chunk": list(ch), "cols": F, "crlf": crlf}
        cells += len(pairs)
        # one column of this copy: count and search, whenever the stride holds every value
        c = int(rng.integers(0, len(flist)))
        texts = [oracle.seek_field(data, index, F, crlf, r, flist[c]) for r in recs]
        if max(len(t) for t in texts) > stride:
            continue
        want = Counter(texts)
        need = pkg.columnar_frequency_scratch_bytes(n)
        scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        ent = torch.zeros((len(want) + 2, 2), dtype=torch.int64, device="cuda:0")
        first_record = recs[0]
        st = pkg.columnar_frequency_device(ctx, cols[c].data_ptr(), lens[c].data_ptr(), n, stride, first_record,
                                           scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
        got, first_of = {}, {}
        for i, t in enumerate(texts):
            first_of.setdefault(t, i)
        for first, cnt in ent[: st.n_distinct].cpu().tolist():
            t = texts[first - first_record]
            if t in got or first_of[t] != first - first_record:
                return cells, {"what": "freq entry", "first": first}
            got[t] = cnt
        if st.n_distinct != len(want) or got != dict(want) or st.truncated or st.overflow:
            return cells, {"what": "freq", "distinct": st.n_distinct, "want": len(want), "n": n, "stride": stride}
        for _ in range(3):
            mode = int(rng.integers(0, 3))
            src = texts[int(rng.integers(0, n))]
            u = rng.random()
            if u < 0.6 and src:
                a = int(rng.integers(0, len(src)))
                b = int(rng.integers(a, min(len(src), a + 256) + 1))
                needle = src[a:b] if mode == 2 else src[: b - a] if mode == 1 else src[:256]
            elif u < 0.8:
                needle = b""
            else:
                needle = bytes(rng.integers(97, 100, size=int(rng.integers(1, 4)), dtype=np.uint8))
            test = (lambda v: v == needle, lambda v: v.startswith(needle), lambda v: needle in v)[mode]
            want_ids = [i for i, t in enumerate(texts) if test(t)]
            bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
            got_n = pkg.columnar_search_device(ctx, cols[c].data_ptr(), lens[c].data_ptr(), n, stride, needle, mode,
                                               bm.data_ptr())
            words = bm.cpu().numpy().view(np.uint64)
            bits = np.unpackbits(words.view(np.uint8), bitorder="little")
            got_ids = np.flatnonzero(bits).tolist()
            if got_n != len(want_ids) or got_ids != want_ids:
                return cells, {"what": "search", "mode": mode, "needle": needle.hex(), "got": got_n, "want": len(want_ids),
                               "stride": stride, "n": n}
    return cells, rowmajor_consumers(rng, ctx, data, index, tape, dbuf, mis, dindex)


def rowmajor_consumers(rng, ctx, data, index, tape, dbuf, mis, dindex):
    """the per-column consumers on the row-major file itself (consumer_kernels.hip): spans, exact count, search"""
    F, crlf = tape.field_cnt, tape.new_line == "CRLF"
    chunks = tape.chunks(min(int(rng.integers(1, 6)), tape.record_cnt - 1))
    f = int(rng.integers(0, F))
    dbytes = dbuf.data_ptr() + mis
    ch = chunks[int(rng.integers(0, len(chunks)))]
    n = ch[3]
    recs = list(oracle.chunk_record_ids(ch, F, crlf))
    texts = [oracle.seek_field(data, index, F, crlf, r, f) for r in recs]
    b = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda:0")
    e = torch.full((n + 1,), -1, dtype=torch.int64, device="cuda:0")
    got_n = pkg.chunk_field_spans_device(dindex.data_ptr(), index.size, F, tape.new_line, ch, f, b.data_ptr(), e.data_ptr())
    bh, eh = b.cpu().tolist(), e.cpu().tolist()
    if got_n != n or bh[n] != -1 or any(data[bh[k]: eh[k]] != texts[k] for k in range(n)):
        return {"what": "spans", "field": f, "chunk": list(ch)}
    all_texts = [oracle.seek_field(data, index, F, crlf, r, f) for c in chunks for r in oracle.chunk_record_ids(c, F, crlf)]
    first_rec = next(iter(oracle.chunk_record_ids(chunks[0], F, crlf)))
    want = Counter(all_texts)
    need = pkg.column_frequency_scratch_bytes(len(all_texts), len(chunks), max([len(t) for t in all_texts] + [1]))
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    ent = torch.zeros((len(want) + 2, 4), dtype=torch.int64, device="cuda:0")
    st = pkg.column_frequency_device(ctx, dbytes, dindex.data_ptr(), index.size, F, tape.new_line, chunks, f,
                                     scratch.data_ptr(), need, ent.data_ptr(), ent.shape[0])
    got, first_of = {}, {}
    for i, t in enumerate(all_texts):
        first_of.setdefault(t, i)
    for first, bb, ee, cnt in ent[: st.n_distinct].cpu().tolist():
        t = data[bb:ee]
        if t in got or first_of.get(t) != first - first_rec:
            return {"what": "row-major freq entry", "first": first, "field": f}
        got[t] = cnt
    if st.n_distinct != len(want) or got != dict(want) or st.n_records != len(all_texts):
        return {"what": "row-major freq", "distinct": st.n_distinct, "want": len(want), "field": f}
    for _ in range(3):
        mode = int(rng.integers(0, 3))
        src = texts[int(rng.integers(0, n))]
        if rng.random() < 0.7 and src:
            a = int(rng.integers(0, len(src)))
            k = int(rng.integers(0, min(len(src) - a, 256) + 1))
            needle = src[a: a + k] if mode == 2 else src[:k] if mode == 1 else src[:256]
        else:
            needle = bytes(rng.integers(97, 100, size=int(rng.integers(0, 4)), dtype=np.uint8))
        test = (lambda v: v == needle, lambda v: v.startswith(needle), lambda v: needle in v)[mode]
        want_ids = [i for i, t in enumerate(texts) if test(t)]
        bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda:0")
        got_n = pkg.column_search_device(ctx, dbytes, len(data), dindex.data_ptr(), index.size, F, tape.new_line, ch, f,
                                         needle, mode, bm.data_ptr())
        bits = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")
        if got_n != len(want_ids) or np.flatnonzero(bits).tolist() != want_ids:
            return {"what": "row-major search", "mode": mode, "needle": needle.hex(), "got": got_n, "want": len(want_ids),
                    "field": f, "chunk": list(ch)}
    return None


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(0)
    t_end = time.time() + seconds
    cases = cells = 0
    bad = []
    t_say = time.time() + 60
    while time.time() < t_end and not bad:
        if time.time() > t_say:   # a line a minute: a silent GPU command is taken for hung after seven
            print(f"# {cases} cases", file=sys.stderr, flush=True)
            t_say = time.time() + 60
        state = rng.bit_generator.state
        k, err = one_case(rng, ctx)
        cells += k
        if err:
            err["case"] = cases
            err["rng_state_before_case"] = state["state"]
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"fuzz_columnar_fail_{seed}_{cases}.csv"), "wb") as f:
                f.write(one_case.last_data)
            bad.append(err)
        cases += 1
    print(json.dumps({"seconds": seconds, "seed": seed, "cases": cases, "cells_checked": cells, "bad": bad}, default=str))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
