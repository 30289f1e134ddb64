# HBM traffic (FETCH_SIZE / WRITE_SIZE passes only) of the kernels whose name contains FILTER -> one line per kernel.
# Usage: bash probes/pmc_traffic.sh OUT.txt FILTER script.py [args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); shift
FILTER=$1; shift
SCRIPT=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
python $SCRIPT "$@" > /dev/null 2>&1
: > $OUT
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_out -o r -- python $SCRIPT "$@" > /tmp/pmc_log.txt 2>&1
  python - "$OUT" "$set" "$FILTER" <<'PY'
import csv, glob, collections, sys
out, sets, filt = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for fn in glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if filt in k:
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
with open(out, 'a') as f:
    f.write(f"# pass: {sets}   (values: SUM over all dispatches of the run / number of dispatches)\n")
    for k in acc:
        f.write(f"  {k[:120]}\n     " + " ".join(f"{c}={v / n[k][c]:.6g}" for c, v in sorted(acc[k].items())) + f"  dispatches={max(n[k].values())}\n")
PY
done
cat $OUT
