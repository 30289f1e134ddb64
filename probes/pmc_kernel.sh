# rocprofv3 PMC passes (separate passes per counter set, --kernel-trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots" / "HBM")
# on the kernels whose name contains FILTER, launched by `python SCRIPT ARGS`.
# Usage: bash probes/pmc_kernel.sh OUT.txt FILTER script.py [args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); shift
FILTER=$1; shift
SCRIPT=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
python $SCRIPT "$@" > /dev/null 2>&1
: > $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_out -o r -- python $SCRIPT "$@" > /tmp/pmc_log.txt 2>&1
  python - "$OUT" "$set" "$FILTER" <<'PY'
import csv, glob, collections, sys
out, sets, filt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
dur = collections.defaultdict(list)
for fn in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if filt in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for fn in glob.glob('/tmp/pmc_out/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if filt in r['Kernel_Name']:
            dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open(out, 'a') as f:
    f.write(f"# pass: {sets}\n")
    if not acc:
        f.write("  (no rows: " + open('/tmp/pmc_log.txt').read()[-300:].replace("\n", " | ") + ")\n")
    for k in acc:
        d = dur.get(k, [0])
        f.write(f"  {k[:120]}\n     " + " ".join(f"{c}={v / n[k][c]:.5g}" for c, v in sorted(acc[k].items()))
                + f"  (avg over {max(n[k].values())} dispatches; kernel time under counters avg {sum(d) / max(len(d), 1) / 1e3:.1f} us)\n")
PY
done
cat $OUT
