# rocprofv3 PMC passes on the production global-attention kernel (32 views x 518^2, f16): separate passes per counter set,
# --kernel-trace only (MI355X_MICROARCH.md sections "rocprofv3 PMC slots" and "HBM").  Usage: bash probes/pmc_attn.sh [static|online] OUT
MODE=${1:-static}; OUT=${2:-gpurun_out/attn_pmc_$MODE.txt}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$OUT")
cd /tmp && export TMPDIR=/tmp
python $REPO/probes/attn_static_only.py $MODE f16 2 > /dev/null
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_out -o r -- python $REPO/probes/attn_static_only.py $MODE f16 3 > /tmp/pmc_log.txt 2>&1
  python - "$OUT" "$set" <<'PY'
import csv, glob, collections, sys
out, sets = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'flash_attn' in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
with open(out, 'a') as f:
    f.write(f"# pass: {sets}\n")
    if not acc:
        f.write("  (no rows: " + open('/tmp/pmc_log.txt').read()[-300:].replace("\n", " | ") + ")\n")
    for k in acc:
        f.write(f"  {k[:110]}\n     " + " ".join(f"{c}={v / n[k][c]:.5g}" for c, v in sorted(acc[k].items())) + f"  (avg over {max(n[k].values())} dispatches)\n")
PY
done
cat $OUT
