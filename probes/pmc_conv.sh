cd /tmp && export TMPDIR=/tmp
python /root/repo/probes/conv_only.py
python /root/repo/probes/conv_only.py 8 296 256 128 3
python /root/repo/probes/conv_only.py 8 518 128 32 3
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pmc_out; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_out -o r -- python /root/repo/probes/conv_only.py > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_out/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'conv_igemm' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(" ".join(f"{k}={v / n[k]:.4g}" for k, v in sorted(acc.items())), flush=True)
PY
done
