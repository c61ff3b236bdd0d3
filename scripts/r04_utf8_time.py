#!/usr/bin/env python3
"""dev tool: time of the UTF-8 validation pass on 1 GiB of text of several scripts (product library).
Latin / Cyrillic / CJK hold none of the bytes the narrowed rules are about; Thai (E0), Hangul (ED) and emoji (F0) do."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
s = torch.cuda.current_stream().cuda_stream
def time_utf8(buf, nbytes):
    r = torch.zeros(2, dtype=torch.int64, device=dev)
    for _ in range(3):
        ctx.utf8_validate_device_async(buf.data_ptr(), nbytes, r.data_ptr(), s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.utf8_validate_device_async(buf.data_ptr(), nbytes, r.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return {"ms": round(ms, 4), "TBps": round(nbytes / ms / 1e9, 3), "first_invalid": int(r[0].item())}
texts = {
    "ascii": "id,name,city,count\n42,Tokyo,Perche no,7\n",
    "latin_mixed": "id,name,città,naïve café\n42,Ünïcödé,Perché no,ok\n",
    "cyrillic": "Съешь же ещё этих мягких французских булок, да выпей чаю\n",
    "cjk": "漢字仮名交じり文",
    "mixed_cjk_ascii": "id,name,città,東京\n",
    "thai_E0": "ภาษาไทยเป็นภาษาที่มีวรรณยุกต์\n",
    "hangul_ED": "한국어는 한반도에서 사용하는 언어이다 훈민정음 힘\n",
    "emoji_F0": "id,emoji\n42,\U0001F680\U0001F600 ok\n",
}
out = {}
big = torch.from_numpy(np.frombuffer(texts["ascii"].encode(), dtype=np.uint8).copy()).to(dev).repeat((8 << 30) // len(texts["ascii"]))
out["ascii_8GiB"] = time_utf8(big, big.numel())
del big
for name, t in texts.items():
    b = torch.from_numpy(np.frombuffer(t.encode(), dtype=np.uint8).copy()).to(dev)
    big = b.repeat((1 << 30) // b.numel())
    out[name] = time_utf8(big, big.numel())
    # a corrupted copy: the byte in the middle of the last repetition's first sequence becomes 0xFF
    bad = big.clone(); pos = big.numel() - b.numel() + 1; bad[pos] = 0xFF
    r = ctx.utf8_validate_device(bad.data_ptr(), bad.numel())
    want = None
    try:
        bad[pos - 1: pos + 63].cpu().numpy().tobytes().decode()   # from the start of the last repetition
    except UnicodeDecodeError as e:
        want = pos - 1 + e.start
    out[name]["corruption_found_at_expected_offset"] = (r == want)
    del big, bad
print(json.dumps(out))
