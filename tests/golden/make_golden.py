#!/usr/bin/env python3
"""Regenerates tests/golden/expected.json.

The reference (a Rust crate) cannot be built in this image (no rustc/cargo, no network), so these
vectors are NOT outputs of the reference binary.  They are derived here by a pure-Python statement
of the reference's stage-1 semantics (src/avx/stage1.rs:384-407: structural = {',', CR, LF} outside
the inclusive prefix-xor of '"'; sentinel 0 first, src/reader.rs:216) and agree with:
  * the reference's own known-answer test  src/reader.rs:325-326
        reader_test01.csv: index[1] == 4, index[last] == 95
  * the vectors derived independently by hand in SURVEY.md §8c (same 17 / 46 / 73 entries)
  * the tape facts the reference's code implies for each file (src/tape.rs:226-273, 315-347).
The three CSV files are the reference's data fixtures (res/*.csv), copied as data.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def stage1(b: bytes):
    idx, inq = [0], False
    for i, ch in enumerate(b):
        if ch == 0x22:
            inq = not inq
        if ch in (0x2C, 0x0A, 0x0D) and not inq:
            idx.append(i)
    return idx


def header(b: bytes):
    end = 0
    while b[end] not in (0x0D, 0x0A):
        end += 1
    crlf = b[end + 1] == 0x0A
    start = 0
    while b[start] in (0xEF, 0xBB, 0xBF):
        start += 1
    names = [s.strip() for s in b[start:end].decode("utf-8").split(",")]
    return {"names": names, "field_cnt": len(names), "new_line": "CRLF" if crlf else "LF",
            "record_offset": end}


def main():
    out = {}
    for name in ("reader_test01.csv", "sample.csv", "sample_rx.csv"):
        b = open(os.path.join(HERE, name), "rb").read()
        idx = stage1(b)
        h = header(b)
        jump = h["field_cnt"] + (1 if h["new_line"] == "CRLF" else 0)
        out[name] = {
            "len": len(b), "index": idx, "header": h, "jump": jump,
            "record_cnt": (len(idx) - 1) // jump, "ragged": (len(idx) - 1) % jump != 0,
        }
    assert out["reader_test01.csv"]["index"][1] == 4 and out["reader_test01.csv"]["index"][-1] == 95
    with open(os.path.join(HERE, "expected.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
