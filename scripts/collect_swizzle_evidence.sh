set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_swz; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the swizzled-window VARIANT library, timed through the same minimal ctypes path as scripts/ab_variants.py
cat > /tmp/swz_run.py <<'PY'
import ctypes as C, sys, torch
L = C.CDLL(sys.argv[1])
ctx = C.c_void_p(); assert L.csvsimd_ctx_create(0, C.byref(ctx)) == 0
cols, width, seed = 1024, 4, 0xC5F00005
n = (1 << 30) // 5120 * 5120
dbuf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
assert L.csvsimd_synth_fill_device(vp(dbuf.data_ptr()), u64(0), u64(n), u32(cols), u32(width), u64(seed), u32(0), None) == 0
cap = n // 5 + 1024
dtape = torch.empty(cap, dtype=torch.int64, device="cuda:0"); dres = torch.zeros(8, dtype=torch.int64, device="cuda:0")
assert L.csvsimd_ctx_reserve(ctx, u64(n)) == 0
ms = C.c_float()
assert L.csvsimd_stage1_time_device(ctx, vp(dbuf.data_ptr()), u64(n), vp(dtape.data_ptr()), u64(cap), vp(dres.data_ptr()), None, 2, 10, C.byref(ms)) == 0
print("kernel ms", ms.value, "count", int(dres[0]))
PY
for lib in variants/swz.so libcsvsimd_hip.so; do
  tag=$(basename $lib .so)
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/$tag -- python3 /tmp/swz_run.py $REPO/csv-simd_amd/csrc/$lib > $OUT/$tag.log 2>&1
  python3 $REPO/scripts/summarise_pmc.py $OUT/$tag $OUT/pmc_$tag.json 1073740800 1717985280 "1024x4_dense 1 GiB ($tag)" > /dev/null
  python3 /tmp/swz_run.py $REPO/csv-simd_amd/csrc/$lib 2>/dev/null | tail -1 > $OUT/time_$tag.txt
  grep "kernel ms" $OUT/$tag.log $OUT/time_$tag.txt
done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
python3 -c "
import json
for t in ('swz','libcsvsimd_hip'):
    d=json.load(open('$OUT/pmc_%s.json'%t)); print(t, {k:v['avg'] for k,v in d['counters'].items()})
"
