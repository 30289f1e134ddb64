# rocprofv3 kernel trace of an arbitrary python command -> per-kernel table.  Usage: bash probes/profile_cmd.sh OUT.txt script.py [args]
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cmd
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cmd -o b -- python "$@" > /tmp/prof_cmd.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for fn in glob.glob('/tmp/prof_cmd/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        rows[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in rows.values())
lines = [f"{'kernel':110s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}"]
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f"{k[:110]:110s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/max(tot,1):6.2f}")
lines.append(f"TOTAL kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in rows.values())} dispatches")
open(sys.argv[1], 'w').write("\n".join(lines) + "\n--- command output (tail) ---\n" + open('/tmp/prof_cmd.log').read()[-2500:])
PY
head -30 "$OUT"
